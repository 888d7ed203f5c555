// bf16 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16) for the bf16-resident path (BASELINE configs[4]).
//
// Every dense projection of the ViLBERT path (vilbert/vilbert.py:285-287, 322, 352, 365, 555-568, 641-644, 831, 864, 906, 968, 1358) and both of
// its backward products with bf16 OPERANDS READ IN PLACE -- nothing is staged, cast or transposed per call:
//   forward          Y  = X  W^T     A = X  [M,K] row-major (contraction contiguous)    B = W  [N,K] row-major (contraction contiguous)
//   input gradient   dX = dY W       A = dY [M,N] row-major (contraction contiguous)    B = W  [N,K]: contraction index = ROW of B
//   weight gradient  dW = dY^T X     A = dY [M,N]: contraction index = ROW of A         B = X  [M,K]: contraction index = ROW of B
// An operand whose contraction index is its row ("k-major") is DMA'd into LDS as it lies in memory, T[k][mn], and the 8 consecutive k one
// lane feeds to the matrix instruction are gathered by two ds_read_b64_tr_b16 (a 16-lane group reads a [4 k][16 mn] block, 8 bytes per lane,
// and lane j receives column j: the hardware transpose), so dY, X and W each exist ONCE in HBM, as bf16, in the layout their producer wrote.
// Accumulation is fp32; the epilogue (bias, erf-GELU with the pre-activation saved, ReLU, x GELU', x ReLU', beta accumulate) runs in fp32 and
// rounds once (RNE) into a bf16 or fp32 C.
//
// Structure (as gemm_dma_kernel of gemm.hip): workgroup tile 256x256 (8 waves of 64x128, one workgroup per CU) or 128x128 (8 waves of 32x64,
// two per CU), 64-deep k-tiles, both operand tiles delivered by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces, no VGPR staging), A into a
// 3-slot ring (two k-tiles ahead), B into a 2-slot ring (one ahead) = all 160 KB of LDS; the main loop is a two-group ping-pong of load and
// matrix phases (four barriers per k-tile, counted vmcnt, see the kernel).  LDS images (the DMA writes lane-linear pieces, so
// the layout is chosen through the per-lane SOURCE address):
//   contraction-contiguous operand  S[m][64]   128-byte rows, 16-byte granule g stored at g ^ ((m >> 1) & 7): conflict-free ds_read_b128
//   k-major operand                 T[k][BMN]  granule g of row k stored at g ^ (4 (k & 3)): the four k rows of a transposing read land on four
//                                              different 64-byte bank groups
// A lane's k slots inside a 64-deep tile are k = 32 h + 8 s + e (h = half-wave, s = matrix instruction 0..3, e = 0..7) for both layouts.
// The last k-tile of a contraction that is not a multiple of 64 is staged through registers with the rows / granules past K zeroed.
// Deterministic split-K (fp32 partials, fixed-order reduce) for the weight gradients; row sums of a k-major A ride on the launch (bias
// gradients, as ytvln_gemm_f32_rowsum).  Shapes the fast path cannot take (unaligned operands) run a small generic kernel.
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace ytvln {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct BfArgs {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; bf16_t* aux;
    int64_t lda, ldb, ldc, ldaux;
    int M, N, K;
    int epilogue;
    float beta;
    int tiles_m, tiles_n, ntiles;
    int splits, kchunk;          // split-K: split s owns k in [s*kchunk, (s+1)*kchunk) (kchunk % 64 == 0); partial tiles go to ws[s][M][N] (fp32)
    float* ws;
    int mnA, mnB;                // clamp extents of the operands in their M / N dimension (rounded up to 8 inside zero padding for k-major operands)
    int kvalidA, kvalidB;        // contraction indices below these are readable: K -- or, for a contraction-contiguous A with zero padding behind K, K rounded up to 8
    float* asum; float* asum_ws; // optional: asum[m] = sum_k A[k][m] of a k-major A
    int stagger_ticks;           // > 0: every second octet of the launch's first round of workgroups starts this many 100 MHz ticks late (see the kernel)
    int wide_stores;             // 1: bf16 C through the LDS-transposed 16-byte stores where the epilogue allows (run-time option GEMM_BF16_WIDE)
    uint32_t* probe; int probe_block, probe_mask;      // diagnostics (ytvln_gemm_bf16_probe): phase time stamps of waves 0 and 4 of this workgroup
};

constexpr int KT = 64;           // k-tile depth in bf16 elements

__device__ __forceinline__ float bf2f(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {          // round to nearest even, NaN stays NaN (v_cvt_pk_bf16_f32 semantics)
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}

template <typename CT> struct Elem;
template <> struct Elem<float> {
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)f2bf(v); }
};

// ---- tiles -----------------------------------------------------------------------------------------------------------------------
// BMN x 64 bf16 = BMN / 8 pieces of 1 KiB; NW waves issue NI pieces each.
template <int BMN, bool KC, int NW>
struct BfTile {
    static constexpr int PIECES = BMN / 8, NI = PIECES / NW;
    static_assert(NI >= 1 && PIECES % NW == 0, "tile too small for the workgroup");
    static constexpr int ROWB = KC ? 128 : BMN * 2;          // bytes per LDS row
    static constexpr int RPP = 1024 / ROWB;                  // LDS rows per piece (KC: 8 rows of m; k-major: 2 or 4 rows of k)
    static constexpr int GPR = ROWB / 16;                    // 16-byte granules per LDS row
    __device__ static __forceinline__ int swz_kc(int m) { return (m >> 1) & 7; }
    // (row, logical granule) this lane fetches for piece c
    __device__ static __forceinline__ void coord(int c, int lane, int& row, int& lg) {
        row = c * RPP + lane / GPR;
        const int pg = lane % GPR;
        lg = KC ? (pg ^ swz_kc(row)) : (pg ^ (4 * (row & 3)));
    }
    // per-lane source pointer of piece c for the tile starting at k0 (no K tail: every row / granule of the tile is real)
    __device__ static __forceinline__ const bf16_t* src(const bf16_t* P, int64_t ld, int MN, int mn0, int k0, int c, int lane) {
        int row, lg;
        coord(c, lane, row, lg);
        if (KC) return P + (int64_t)min(mn0 + row, MN - 1) * ld + k0 + 8 * lg;
        return P + (int64_t)(k0 + row) * ld + min(mn0 + 8 * lg, MN - 8);
    }
    __device__ static __forceinline__ int64_t step(int64_t ld) { return KC ? KT : KT * ld; }
    // K tail: the same 16 bytes through a register, zero where the contraction index is >= kvalid
    __device__ static __forceinline__ uint4 tail(const bf16_t* P, int64_t ld, int MN, int mn0, int k0, int c, int lane, int kvalid) {
        int row, lg;
        coord(c, lane, row, lg);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (KC) {
            if (k0 + 8 * lg < kvalid) v = *reinterpret_cast<const uint4*>(P + (int64_t)min(mn0 + row, MN - 1) * ld + k0 + 8 * lg);
        } else {
            if (k0 + row < kvalid) v = *reinterpret_cast<const uint4*>(P + (int64_t)(k0 + row) * ld + min(mn0 + 8 * lg, MN - 8));
        }
        return v;
    }
};

// Per-lane fragment addressing.  KC: byte offset of the lane's row, the granule is (4 h + s) ^ swz -> off[i] = row * 128 and the swizzle key.
// k-major: byte offset of (row 32 h + r, this lane's 4 columns of sub-tile i) with the swizzle applied (r = (lane % 16) / 4 is the row's low
// two bits for every read: 32 h + 8 s + 4 t is a multiple of 4), the (s, t) part is an immediate (8 s + 4 t) * ROWB.
template <int BMN, bool KC, int NSUB>
struct BfFrag {
    int off[NSUB];
    int key[NSUB];          // KC only
    __device__ __forceinline__ void init(int w0, int lane) {
        const int l31 = lane & 31, h = lane >> 5;
        if (KC) {
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                const int row = w0 + 32 * i + l31;
                off[i] = row * 128;
                key[i] = (row >> 1) & 7;
            }
        } else {
            const int r = (lane & 15) >> 2, c4 = lane & 3, blk = (lane >> 4) & 1;
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                const int col = w0 + 32 * i + 16 * blk + 4 * c4;
                off[i] = (32 * h + r) * (BMN * 2) + (((col >> 3) ^ (4 * r)) << 4) + ((col & 7) << 1);
                key[i] = 0;
            }
        }
    }
    // the 8 bf16 this lane feeds to matrix instruction s of the k-tile at S (sub-tile i)
    __device__ __forceinline__ bf16x8 get(const char* __restrict__ S, int i, int h, int s) const {
        if (KC) {
            return *reinterpret_cast<const bf16x8*>(S + off[i] + (((4 * h + s) ^ key[i]) << 4));
        } else {
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const char* p = S + off[i] + (8 * s) * (BMN * 2);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_ptr_t)(const_cast<char*>(p)));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_ptr_t)(const_cast<char*>(p + 4 * (BMN * 2))));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        }
    }
};

// ---- epilogue ------------------------------------------------------------------------------------------------------------------------
// lane owns column l31 of each 32x32 tile, rows (r & 3) + 8 (r >> 2) + 4 h.  Epilogue kind and the "tile inside the matrix" test are resolved
// once per wave (uniform branches around specialised loops), as in gemm.hip.
template <int TM, int TN, int EPI, bool INTERIOR, typename CT>
__device__ __forceinline__ void bf_epilogue_body(const BfArgs& g, f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int half) {
    const bool has_beta = g.beta != 0.f;
    CT* const C = reinterpret_cast<CT*>(g.C);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + 32 * j + l31;
        if (!INTERIOR && col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = row0 + 32 * i + 4 * half;
            CT* cp0 = C + (int64_t)rbase * g.ldc + col;
            bf16_t* xp0 = (EPI != YTVLN_EPI_NONE && EPI != YTVLN_EPI_RELU) ? g.aux + (int64_t)rbase * g.ldaux + col : nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float ax[4], old[4];
                if (EPI == YTVLN_EPI_MUL_DGELU || EPI == YTVLN_EPI_MUL_DRELU) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ax[u] = (INTERIOR || rbase + 8 * q + u < g.M) ? bf2f(xp0[(int64_t)(8 * q + u) * g.ldaux]) : 0.f;
                }
                if (has_beta) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) old[u] = (INTERIOR || rbase + 8 * q + u < g.M) ? Elem<CT>::ld(cp0 + (int64_t)(8 * q + u) * g.ldc) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int dr = 8 * q + u;
                    if (!INTERIOR && rbase + dr >= g.M) continue;
                    float v = acc[i][j][4 * q + u] + bv;
                    if (EPI == YTVLN_EPI_GELU) {
                        if (g.aux) xp0[(int64_t)dr * g.ldaux] = (bf16_t)f2bf(v);
                        v = gelu_erf_b(v);
                    } else if (EPI == YTVLN_EPI_RELU) {
                        v = fmaxf(v, 0.f);
                    } else if (EPI == YTVLN_EPI_MUL_DGELU) {
                        v *= dgelu_erf_b(ax[u]);
                    } else if (EPI == YTVLN_EPI_MUL_DRELU) {
                        v = ax[u] > 0.f ? v : 0.f;
                    }
                    if (has_beta) v += g.beta * old[u];
                    Elem<CT>::st(cp0 + (int64_t)dr * g.ldc, v);
                    if (EPI == YTVLN_EPI_GELU || EPI == YTVLN_EPI_MUL_DGELU) __builtin_amdgcn_sched_barrier(0);          // (see bf_epilogue_interior)
                }
            }
        }
    }
}

// Interior tiles: the same arithmetic with scalar row bases + ONE per-lane byte offset per matrix and hand-counted loads (gemm_tiles.h
// epilogue_interior has the story: the per-lane 64-bit pointers of the generic body made hipcc spill beside the 128 accumulators of a 256x256
// tile, and every scratch reload waited for all stores in flight).  C is bf16 (2-byte stores) or fp32; aux is bf16.
template <int TM, int TN, int EPI, bool BETA, typename CT>
__device__ __forceinline__ void bf_epilogue_interior(const BfArgs& g, f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int half) {
    constexpr bool AUX_IN = EPI == YTVLN_EPI_MUL_DGELU || EPI == YTVLN_EPI_MUL_DRELU;
    constexpr bool C16 = std::is_same<CT, bf16_t>::value;
    constexpr int NL = (AUX_IN ? 4 : 0) + (BETA ? 4 : 0), NST = 4, G = TN * TM * 4;
    constexpr int CB = (int)sizeof(CT);
    const uint32_t vc = (uint32_t)(((int64_t)(4 * half) * g.ldc + l31) * CB);
    const uint32_t vx = (uint32_t)(((int64_t)(4 * half) * g.ldaux + l31) * 2);
    const char* const cb = scalar_ptr(reinterpret_cast<const char*>(reinterpret_cast<CT*>(g.C) + (int64_t)row0 * g.ldc + col0));
    const char* const xb = (EPI != YTVLN_EPI_NONE && EPI != YTVLN_EPI_RELU && g.aux) ? scalar_ptr(reinterpret_cast<const char*>(g.aux + (int64_t)row0 * g.ldaux + col0)) : nullptr;
    const int64_t cs = g.ldc * CB, xs = g.ldaux * 2;
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = g.bias ? g.bias[col0 + 32 * j + l31] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bv[j]));
    float ax[2][4] = {}, old[2][4] = {};
    auto row_c = [&](int gi) { const int j = gi / (TM * 4), i = (gi / 4) % TM, q = gi % 4; const char* p = cb + (int64_t)(32 * i + 8 * q) * cs + 32 * CB * j; asm volatile("" : "+s"(p)); return p; };
    auto row_x = [&](int gi) { const int j = gi / (TM * 4), i = (gi / 4) % TM, q = gi % 4; const char* p = xb + (int64_t)(32 * i + 8 * q) * xs + 64 * j; asm volatile("" : "+s"(p)); return p; };
    auto loads = [&](auto gc) {
        constexpr int gi = decltype(gc)::value;
        if constexpr (AUX_IN) {
            const char* const px = row_x(gi);
#pragma unroll
            for (int u = 0; u < 4; ++u) epi_load_ushort(ax[gi & 1][u], vx, px + u * xs);
        }
        if constexpr (BETA) {
            const char* const pc = row_c(gi);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (C16) epi_load_ushort(old[gi & 1][u], vc, pc + u * cs);
                else epi_load(old[gi & 1][u], vc, pc + u * cs);
            }
        }
    };
    if constexpr (NL > 0) loads(std::integral_constant<int, 0>{});
    static_for<G>([&](auto gc) {
        constexpr int gi = decltype(gc)::value;
        constexpr int j = gi / (TM * 4), i = (gi / 4) % TM, q = gi % 4;
        if constexpr (NL > 0) {
            if constexpr (gi + 1 < G) loads(std::integral_constant<int, gi + 1>{});
            epi_wait<(gi > 0 ? NST : 0) + (gi + 1 < G ? NL : 0)>(ax[gi & 1], old[gi & 1]);
        }
        const char* const pc = row_c(gi);
        const char* const px = (EPI == YTVLN_EPI_GELU && xb) ? row_x(gi) : nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = acc[i][j][4 * q + u] + bv[j];
            if (EPI == YTVLN_EPI_GELU) {
                if (px) epi_store_short(vx, f2bf(v), px + u * xs);
                v = gelu_erf_b(v);
            } else if (EPI == YTVLN_EPI_RELU) {
                v = fmaxf(v, 0.f);
            } else if (EPI == YTVLN_EPI_MUL_DGELU) {
                v *= dgelu_erf_b(bf2f(__float_as_uint(ax[gi & 1][u])));
            } else if (EPI == YTVLN_EPI_MUL_DRELU) {
                v = bf2f(__float_as_uint(ax[gi & 1][u])) > 0.f ? v : 0.f;
            }
            if (BETA) v += g.beta * (C16 ? bf2f(__float_as_uint(old[gi & 1][u])) : old[gi & 1][u]);
            if constexpr (C16) epi_store_short(vc, f2bf(v), pc + u * cs);
            else epi_store(vc, v, pc + u * cs);
            // one value's Phi at a time: interleaved, the four evaluations of a group (eight live temporaries each) spill beside the 128 accumulators
            if constexpr (EPI == YTVLN_EPI_GELU || EPI == YTVLN_EPI_MUL_DGELU) __builtin_amdgcn_sched_barrier(0);
        }
    });
}

template <int TM, int TN, typename CT>
__device__ __forceinline__ void bf_epilogue(const BfArgs& g, f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int half, int split) {
    const bool interior = row0 + 32 * TM <= g.M && col0 + 32 * TN <= g.N;
    if (g.splits > 1) {      // raw fp32 partial sums; bias / beta / rounding are applied by bf_splitk_reduce_kernel in a fixed order
        float* w = g.ws + (int64_t)split * g.M * g.N;
        if (interior) {          // scalar row bases + one per-lane offset
            const uint32_t vw = (uint32_t)(((int64_t)(4 * half) * g.N + l31) * 4);
            const char* const wb = scalar_ptr(reinterpret_cast<const char*>(w + (int64_t)row0 * g.N + col0));
            const int64_t wsb = (int64_t)g.N * 4;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const char* p = wb + (int64_t)(32 * i + 8 * q) * wsb + 128 * j;
                        asm volatile("" : "+s"(p));
#pragma unroll
                        for (int u = 0; u < 4; ++u) epi_store(vw, acc[i][j][4 * q + u], p + u * wsb);
                    }
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + 32 * j + l31;
            if (!interior && col >= g.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = row0 + 32 * i + 4 * half;
                float* wp = w + (int64_t)rbase * g.N + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (interior || rbase + dr < g.M) wp[(int64_t)dr * g.N] = acc[i][j][r];
                }
            }
        }
        return;
    }
#define YT_BF_EPI(E)                                                                              \
    case E:                                                                                       \
        if (interior) { if (g.beta != 0.f) bf_epilogue_interior<TM, TN, E, true, CT>(g, acc, row0, col0, l31, half); else bf_epilogue_interior<TM, TN, E, false, CT>(g, acc, row0, col0, l31, half); } \
        else bf_epilogue_body<TM, TN, E, false, CT>(g, acc, row0, col0, l31, half);                \
        break;
    switch (g.epilogue) {
        YT_BF_EPI(YTVLN_EPI_GELU)
        YT_BF_EPI(YTVLN_EPI_RELU)
        YT_BF_EPI(YTVLN_EPI_MUL_DGELU)
        YT_BF_EPI(YTVLN_EPI_MUL_DRELU)
        default:
            if (interior) { if (g.beta != 0.f) bf_epilogue_interior<TM, TN, YTVLN_EPI_NONE, true, CT>(g, acc, row0, col0, l31, half); else bf_epilogue_interior<TM, TN, YTVLN_EPI_NONE, false, CT>(g, acc, row0, col0, l31, half); }
            else bf_epilogue_body<TM, TN, YTVLN_EPI_NONE, false, CT>(g, acc, row0, col0, l31, half);
            break;
    }
#undef YT_BF_EPI
}

// ---- wide stores of a bf16 C ----------------------------------------------------------------------------------------------------------
// The matrix instruction leaves a lane with 4 consecutive ROWS of one column per register group, i.e. 2-byte stores (128 per wave on a 64x128
// sub-tile; measured: the store tail is a third of a tile's fixed cost).  For tiles inside the matrix and epilogues that read no matrix (plain /
// bias, ReLU, GELU with or without the saved pre-activation) each wave instead transposes one 32-row block at a time through its own slice of the
// -- by then idle -- LDS: packed bf16 column segments written with ds_write_b64 (column stride 72 bytes: conflict-free), read back by
// ds_read_b64_tr_b16 as 8 consecutive columns of a row, stored 16 bytes per lane: 2 TN store instructions per block instead of 16 TN.
constexpr int BF_TCS = 72;          // bytes per column of the transposing image: 32 rows x 2 B + 8
template <int TN>
__device__ __forceinline__ constexpr int bf_wide_bytes() { return 32 * TN * BF_TCS; }          // LDS per wave
// MODE 0: v + bias (the plain product), 1: relu.  Write half: one 32-row block into its image at tb.
template <int TN> struct BfBlk { f32x16 v[TN]; };          // the accumulators of one 32-row block of a wave's tile
template <int TN, int MODE>
__device__ __forceinline__ void bf_wide_write(const BfBlk<TN>& blkw, const float (&bv)[TN], char* __restrict__ tb, int lane) {
    const f32x16 (&blk)[TN] = blkw.v;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v = blk[j][4 * q + u] + bv[j];
                if (MODE == 1) v = fmaxf(v, 0.f);
                h[u] = f2bf(v);
            }
            *reinterpret_cast<uint2*>(tb + (32 * j + l31) * BF_TCS + (8 * q + 4 * half) * 2) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        }
}
// Read half: lane (group gq = lane >> 4, jj = lane & 15) ends up with row 16 rh + jj, columns 32 cb + 8 gq .. + 7; for the transposing read it SUPPLIES
// the address of T[column 32 cb + 8 gq + 4 u + rr][rows 16 rh + 4 c4 ..] (rr = jj >> 2, c4 = jj & 3)
template <int TN>
__device__ __forceinline__ void bf_wide_store(const char* __restrict__ tb, bf16_t* __restrict__ dst, int64_t ld, int row0, int col0, int lane) {
    const int gq = lane >> 4, jj = lane & 15, rr = jj >> 2, c4 = jj & 3;
#pragma unroll
    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int cb = 0; cb < TN; ++cb) {
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const char* p0 = tb + (32 * cb + 8 * gq + rr) * BF_TCS + (16 * rh + 4 * c4) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_ptr_t)(const_cast<char*>(p0)));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_ptr_t)(const_cast<char*>(p0 + 4 * BF_TCS)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<s16x8*>(dst + (int64_t)(row0 + 16 * rh + jj) * ld + col0 + 32 * cb + 8 * gq) = v;
        }
}
// may this wave's TM x TN block grid at (row0, col0) take the wide path?  (wave-uniform)
template <int TM, int TN>
__device__ __forceinline__ bool bf_wide_ok(const BfArgs& g, int row0, int col0) {
    // (plain / bias and ReLU; the GELU epilogue keeps the 2-byte-store path: its branch-free Phi spills beside the block images of this one, and two
    // different erf evaluations in the two paths would end the bit-identity between them that tests/test_bf16_gpu.py holds them to)
    return g.wide_stores && g.beta == 0.f && (g.epilogue == YTVLN_EPI_NONE || g.epilogue == YTVLN_EPI_RELU) && g.splits == 1 &&
           row0 + 32 * TM <= g.M && col0 + 32 * TN <= g.N && (g.ldc % 8) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0;
}
// The whole TM x TN block grid of a wave, EVERY block in its own LDS image (TM x bf_wide_bytes<TN>() per wave: 147 KB per workgroup on the
// 256x256 tiles), so that the LDS round trip is paid once per pass and not once per block: all blocks written, one wait, all blocks stored.
// `get(i)` hands out block i's accumulators (a reference, or a copy read out of the accumulation registers).
template <int TM, int TN, int MODE, class Get>
__device__ __forceinline__ void bf_wide_pass(Get&& get, const float (&bv)[TN], char* __restrict__ tb, bf16_t* __restrict__ dst, int64_t ld, int row0, int col0,
                                             int lane) {
    static_for<TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        bf_wide_write<TN, MODE>(get(ic), bv, tb + i * bf_wide_bytes<TN>(), lane);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TM; ++i) bf_wide_store<TN>(tb + i * bf_wide_bytes<TN>(), dst, ld, row0 + 32 * i, col0, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads have returned before a next pass overwrites the images
}
template <int TM, int TN, class Get>
__device__ __forceinline__ void bf_wide_tile(const BfArgs& g, Get&& get, char* __restrict__ tb, int row0, int col0, int lane) {
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = g.bias ? g.bias[col0 + 32 * j + (lane & 31)] : 0.f;
    bf16_t* const Cp = reinterpret_cast<bf16_t*>(g.C);
    if (g.epilogue == YTVLN_EPI_RELU) {
        bf_wide_pass<TM, TN, 1>(get, bv, tb, Cp, g.ldc, row0, col0, lane);
    } else {
        bf_wide_pass<TM, TN, 0>(get, bv, tb, Cp, g.ldc, row0, col0, lane);
    }
}

// workgroup id -> (tile row, tile column, split): the XCD-aware orders of gemm.hip (each XCD gets a contiguous run of the group-major tile
// order; with split-K a contiguous run of the split-major (split, tile) order, so neighbouring tiles of ONE k range share an L2)
struct BfCoord { int m, n, split; };
__device__ __forceinline__ BfCoord bf_decode(int bid, int tiles_m, int tiles_n, int splits) {
    const int ntiles = tiles_m * tiles_n;
    const int id = xcd_remap(bid, ntiles * splits);
    BfCoord c;
    c.split = id / ntiles;
    const int t = id - c.split * ntiles;
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int gidx = t / per_group, first_m = gidx * GROUP_M;
    const int rows = min(tiles_m - first_m, GROUP_M);
    const int r = t - gidx * per_group;
    c.m = first_m + r % rows;
    c.n = r / rows;
    return c;
}

#if defined(YT_BF16_SHARED_ONLY)
}  // namespace ytvln
#else
// ---- main kernel -----------------------------------------------------------------------------------------------------------------------
// FORM: where the LDS-DMA of the next operand tiles is issued.  0: inside the load phases (L01: B(kt+1), L23: A(kt+2)), as rounds 4-5 shipped it;
// 1: A(kt+2) between the matrix instructions of M23; 2: also B(kt+1) between those of M01 -- the load phases then hold fragment reads only and
// fit under the partner group's 16 matrix instructions (a DMA piece costs 60-185 cycles of issue in a phase that also carries 12 LDS reads).
template <int BM, int BN, bool A_KC, bool B_KC, int NW, int WPS, typename CT, int FORM = 0, bool PROBE = false>
__global__ __launch_bounds__(NW * 64, WPS) void gemm_bf16_kernel(const BfArgs g) {
    using TA = BfTile<BM, A_KC, NW>;
    using TB = BfTile<BN, B_KC, NW>;
    constexpr int WM = NW / 2;                          // waves along m (x 2 along n)
    constexpr int TM = BM / WM / 32, TN = BN / 64;
    constexpr int SA = BM * KT * 2, SB = BN * KT * 2;                       // bytes per operand tile
    // LDS rings: THREE slots for A, TWO for B -- exactly the 160 KB of a CU for one 256x256 workgroup (two 128x128 ones).  A (the activation /
    // gradient matrix, streamed from HBM once) is fetched TWO k-tiles ahead, B (the weight panel every workgroup of a tile column re-reads: L2)
    // one; with two slots each the next tile had a single k-tile of matrix time (~1 us) to arrive and the wait in front of the barrier drained it.
    constexpr int NSA = 3, NSB = 2;
    __shared__ __attribute__((aligned(16))) char smem[NSA * SA + NSB * SB];  // ONE shared object (a second one de-pipelines the DMA)
    char* const ringA = smem;
    char* const ringB = smem + NSA * SA;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * (BM / WM), wn0 = (wave & 1) * (BN / 2);
    const BfCoord tc = bf_decode(blockIdx.x, g.tiles_m, g.tiles_n, g.splits);
    const int m0 = tc.m * BM, n0 = tc.n * BN;
    const int kbeg = tc.split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + KT - 1) / KT;
    const int nfull = (kend - kbeg) / KT;                     // whole k-tiles: LDS-DMA; a last partial one goes through registers
    if (g.stagger_ticks > 0 && blockIdx.x < 256u && ((blockIdx.x >> 3) & 1)) {
        // De-synchronise the chip: with one workgroup per CU every CU of a many-round launch reaches its prologue (cold A panels from HBM) and
        // its store burst at the same moment, round after round.  Half of the FIRST round starts half a tile late; later workgroups inherit the
        // offset from the CU they replace, so one half's fixed costs run under the other half's main loop.
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)g.stagger_ticks) __builtin_amdgcn_s_sleep(16);
    }

    const bf16_t* pa[TA::NI];
    const bf16_t* pb[TB::NI];
#pragma unroll
    for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg, wave * TA::NI + i, lane);
#pragma unroll
    for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg, wave * TB::NI + i, lane);
    const int64_t sa = TA::step(g.lda), sb = TB::step(g.ldb);

    BfFrag<BM, A_KC, TM> fa;
    BfFrag<BN, B_KC, TN> fb;
    fa.init(wm0, lane);
    fb.init(wn0, lane);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_asum = !A_KC && g.asum != nullptr && tc.n == 0 && (wave & 1) == 0;
    float asum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;

    int slotA_in = 0, slotB_in = 0;          // ring slots the next issues fill
    auto issueA = [&](int kt) {
        char* As = ringA + slotA_in * SA;
        slotA_in = slotA_in + 1 == NSA ? 0 : slotA_in + 1;
        if (kt < nfull) {
#pragma unroll
            for (int i = 0; i < TA::NI; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[i], (lds_ptr_t)(As + (wave * TA::NI + i) * 1024), 16, 0, 0);
                pa[i] += sa;
            }
        } else {          // rare: the tile crosses K -- the same pieces through registers, zero past the end (the waits of the loop cover the ds_writes)
            const int k0 = kbeg + kt * KT;
#pragma unroll
            for (int i = 0; i < TA::NI; ++i)
                *reinterpret_cast<uint4*>(As + (wave * TA::NI + i) * 1024 + 16 * lane) = TA::tail(g.A, g.lda, g.mnA, m0, k0, wave * TA::NI + i, lane, g.kvalidA);
        }
    };
    auto issueB = [&](int kt) {
        char* Bs = ringB + slotB_in * SB;
        slotB_in = slotB_in + 1 == NSB ? 0 : slotB_in + 1;
        if (kt < nfull) {
#pragma unroll
            for (int i = 0; i < TB::NI; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[i], (lds_ptr_t)(Bs + (wave * TB::NI + i) * 1024), 16, 0, 0);
                pb[i] += sb;
            }
        } else {
            const int k0 = kbeg + kt * KT;
#pragma unroll
            for (int i = 0; i < TB::NI; ++i)
                *reinterpret_cast<uint4*>(Bs + (wave * TB::NI + i) * 1024 + 16 * lane) = TB::tail(g.B, g.ldb, g.mnB, n0, k0, wave * TB::NI + i, lane, g.kvalidB);
        }
    };

    // Main loop = a two-group PING-PONG (guide: 8-phase GEMM template).  The 8 waves are two groups of four (one wave per SIMD each); a k-tile
    // is four phases, each closed by a workgroup barrier:
    //     L01  read the fragments of matrix steps s = 0, 1 from LDS; issue the DMA of B(kt+1)            M01  16 (TM TN x 2) matrix instructions
    //     L23  read s = 2, 3; issue the DMA of A(kt+2); wait for this wave's pieces of tile kt+1          M23  16 matrix instructions
    // and group 1 runs ONE barrier behind group 0 (it takes an extra barrier before the loop, group 0 one after it), so whenever one wave of a
    // SIMD is in a load phase -- LDS-DMA issue (~100 cycles a piece), 12 LDS reads, waits -- the other one is in a matrix phase: the two waves of
    // a SIMD no longer stall on the same thing at the same time (in lock-step the matrix pipe idled through every DMA-issue / read phase).
    // Hazards (slot = interval between two barrier releases; group g runs phase p in slot p + g):
    //   * a ring slot is refilled at least one full slot after the other group's last read of it (A: 3 slots, refilled in L23(kt) with tile kt+2,
    //     last read in L23(kt-1); B: 2 slots, refilled in L01(kt) with tile kt+1, last read in L23(kt-1)), and every read is retired (lgkmcnt(0))
    //     before the barrier that closes its phase;
    //   * every wave waits for its own pieces of tile kt+1 in L23(kt); group 0 first reads that tile two barriers later, group 1 three.
    //   vector memory retires in order and B(kt+1) is issued before A(kt+2): "at most NI_A outstanding" = tile kt+1 complete, A(kt+2) in flight.
    const int grp = wave >> 2;                                   // waves 0-3 / 4-7 (wave-uniform SGPR)
    // PROBE: waves 0 and 4 of workgroup probe_block stamp s_memtime (shader cycles, low 32 bits) at the start and end of each phase's own
    // work for k-tiles 8..15 -- 8 stamps per k-tile into the 64 lanes of one register, written out at the end: [own work ends | barrier released]
    uint32_t ts = 0;
    const bool probing = PROBE && g.probe != nullptr && (int)blockIdx.x == g.probe_block && (wave & 3) == 0;
    auto stamp = [&](int kt, int p) __attribute__((always_inline)) {
        if constexpr (PROBE) {
            if (probing && kt >= 8 && kt < 16 && ((g.probe_mask >> p) & 1)) {
                const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime();
                ts = lane == (kt - 8) * 8 + p ? now : ts;
            }
        }
    };
    bf16x8 fra[2][TM], frb[2][TN];
    auto read_frags = [&](const char* __restrict__ As, const char* __restrict__ Bs, int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fra[u][i] = fa.get(As, i, half, s0 + u);
#pragma unroll
            for (int j = 0; j < TN; ++j) frb[u][j] = fb.get(Bs, j, half, s0 + u);
        }
    };
    // the DMA pieces of one operand tile issued between the matrix instructions of a phase: piece i after matrix instruction 4 i + 1
    auto matrix_phase_dma = [&](auto which, int kt_in, bool live) __attribute__((always_inline)) {
        constexpr bool IS_A = decltype(which)::value;
        constexpr int NIP = IS_A ? TA::NI : TB::NI;
        constexpr int NMFMA = 2 * TM * TN;
        char* S = nullptr;
        if (live) {
            if constexpr (IS_A) { S = ringA + slotA_in * SA; slotA_in = slotA_in + 1 == NSA ? 0 : slotA_in + 1; }
            else { S = ringB + slotB_in * SB; slotB_in = slotB_in + 1 == NSB ? 0 : slotB_in + 1; }
        }
        const bool dma = live && kt_in < nfull;
        if (live && !dma) {          // rare: the tile crosses K -- through registers, zero past the end
            const int k0 = kbeg + kt_in * KT;
#pragma unroll
            for (int i = 0; i < NIP; ++i) {
                if constexpr (IS_A) *reinterpret_cast<uint4*>(S + (wave * NIP + i) * 1024 + 16 * lane) = TA::tail(g.A, g.lda, g.mnA, m0, k0, wave * NIP + i, lane, g.kvalidA);
                else *reinterpret_cast<uint4*>(S + (wave * NIP + i) * 1024 + 16 * lane) = TB::tail(g.B, g.ldb, g.mnB, n0, k0, wave * NIP + i, lane, g.kvalidB);
            }
        }
        __builtin_amdgcn_s_setprio(1);
        static_for<NMFMA>([&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            constexpr int u = idx / (TM * TN), i = (idx / TN) % TM, j = idx % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[u][i], frb[u][j], acc[i][j], 0, 0, 0);
            static_for<NIP>([&](auto pc) {          // piece pi behind matrix instruction floor(pi * NMFMA / NIP): evenly spread over the phase
                constexpr int pi = decltype(pc)::value;
                if constexpr ((pi * NMFMA) / NIP == idx) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (dma) {
                        if constexpr (IS_A) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[pi], (lds_ptr_t)(S + (wave * NIP + pi) * 1024), 16, 0, 0); pa[pi] += sa; }
                        else { __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[pi], (lds_ptr_t)(S + (wave * NIP + pi) * 1024), 16, 0, 0); pb[pi] += sb; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
        __builtin_amdgcn_s_setprio(0);
    };
    auto asum_phase = [&]() __attribute__((always_inline)) {
        if constexpr (!A_KC) {
            if (do_asum) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float t = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) t += (float)fra[u][i][e];
                        asum[i] += t;
                    }
            }
        }
    };
    auto matrix_phase = [&]() __attribute__((always_inline)) {
        if constexpr (!A_KC) {
            if (do_asum) {          // wave-uniform: first tile column, first wave column (row sums of A: the bias gradient)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float t = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) t += (float)fra[u][i][e];
                        asum[i] += t;
                    }
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[u][i], frb[u][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    if (nk > 0) { issueA(0); issueB(0); }
    if (nk > 1) issueA(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(TA::NI) : "memory");          // A(0), B(0) landed; A(1) may be in flight
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                  // the stagger: group 1 one slot behind from here on
    int slotA_out = 0, slotB_out = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* As = ringA + slotA_out * SA;
        const char* Bs = ringB + slotB_out * SB;
        slotA_out = slotA_out + 1 == NSA ? 0 : slotA_out + 1;
        slotB_out = slotB_out + 1 == NSB ? 0 : slotB_out + 1;
        // ---- L01
        read_frags(As, Bs, 0);
        if constexpr (FORM < 2) { if (kt + 1 < nk) issueB(kt + 1); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stamp(kt, 0);
        __builtin_amdgcn_s_barrier();
        stamp(kt, 1);
        // ---- M01
        if constexpr (FORM == 2) {
            asum_phase();
            matrix_phase_dma(std::false_type{}, kt + 1, kt + 1 < nk);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the K-tail form of the issue writes LDS through registers)
        } else {
            matrix_phase();
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(kt, 2);
        __builtin_amdgcn_s_barrier();
        stamp(kt, 3);
        // ---- L23
        read_frags(As, Bs, 2);
        if constexpr (FORM == 0) {
            if (kt + 2 < nk) {
                issueA(kt + 2);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(TA::NI) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        } else {          // everything issued so far -- A(kt+1) in M23(kt-1), B(kt+1) in L01 / M01(kt) -- has landed before the barrier below
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(kt, 4);
        __builtin_amdgcn_s_barrier();
        stamp(kt, 5);
        // ---- M23
        if constexpr (FORM >= 1) {
            asum_phase();
            matrix_phase_dma(std::true_type{}, kt + 2, kt + 2 < nk);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            matrix_phase();
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(kt, 6);
        __builtin_amdgcn_s_barrier();
        stamp(kt, 7);
    }
    if constexpr (PROBE) {
        if (probing) g.probe[grp * 64 + lane] = ts;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                  // (every wave executes the same number of barriers)
    if constexpr (!A_KC) {
        if (do_asum) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v = asum[i] + __shfl_xor(asum[i], 32, 64);        // the two half-waves own disjoint k
                const int row = m0 + wm0 + 32 * i + l31;
                if (half == 0 && row < g.M) {
                    if (g.splits > 1) g.asum_ws[(int64_t)tc.split * g.M + row] = v;
                    else g.asum[row] = v;
                }
            }
        }
    }
    if constexpr (std::is_same<CT, bf16_t>::value) {
        if (bf_wide_ok<TM, TN>(g, m0 + wm0, n0 + wn0)) {          // (behind the loop's last barrier nobody reads the operand rings any more)
            char* const tb = smem + wave * (TM * bf_wide_bytes<TN>());
            bf_wide_tile<TM, TN>(g, [&](auto ic) {
                BfBlk<TN> b;
#pragma unroll
                for (int j = 0; j < TN; ++j) b.v[j] = acc[decltype(ic)::value][j];
                return b;
            }, tb, m0 + wm0, n0 + wn0, lane);
            return;
        }
    }
    bf_epilogue<TM, TN, CT>(g, acc, m0 + wm0, n0 + wn0, l31, half, tc.split);
}

// One launch function per (output type, main-loop FORM); forms 1 and 2 are instantiated in their own translation units (gemm_bf16_f1.hip,
// gemm_bf16_f2.hip include this file with YT_BF16_FORM_TU), so the three sets of kernels build in parallel.
template <typename CT, int FORM>
void bf_launch_form(const BfArgs& g, int big, int transA, int transB, hipStream_t s) {
    const dim3 grid((unsigned)(g.ntiles * g.splits)), blk(512);
#define YT_BF(BMV, WPSV)                                                                                                                    \
    do {                                                                                                                                    \
        if (!transA && transB) hipLaunchKernelGGL((gemm_bf16_kernel<BMV, BMV, true, true, 8, WPSV, CT, FORM>), grid, blk, 0, s, g);          \
        else if (!transA && !transB) hipLaunchKernelGGL((gemm_bf16_kernel<BMV, BMV, true, false, 8, WPSV, CT, FORM>), grid, blk, 0, s, g);   \
        else if (transA && !transB) hipLaunchKernelGGL((gemm_bf16_kernel<BMV, BMV, false, false, 8, WPSV, CT, FORM>), grid, blk, 0, s, g);   \
        else hipLaunchKernelGGL((gemm_bf16_kernel<BMV, BMV, false, true, 8, WPSV, CT, FORM>), grid, blk, 0, s, g);                           \
    } while (0)
    if (big) YT_BF(256, 2);
    else YT_BF(128, 4);
#undef YT_BF
}

#if defined(YT_BF16_FORM_TU)
template void bf_launch_form<float, YT_BF16_FORM_TU>(const BfArgs&, int, int, int, hipStream_t);
template void bf_launch_form<bf16_t, YT_BF16_FORM_TU>(const BfArgs&, int, int, int, hipStream_t);
}  // namespace ytvln
#else
extern template void bf_launch_form<float, 1>(const BfArgs&, int, int, int, hipStream_t);
extern template void bf_launch_form<bf16_t, 1>(const BfArgs&, int, int, int, hipStream_t);
extern template void bf_launch_form<float, 2>(const BfArgs&, int, int, int, hipStream_t);
extern template void bf_launch_form<bf16_t, 2>(const BfArgs&, int, int, int, hipStream_t);
// 32-deep k-tiles in five-slot rings, three tiles in flight (gemm_bf16_h.hip)
template <typename CT> void bf_launch_h(const BfArgs& g, int big, int transA, int transB, hipStream_t s);
extern template void bf_launch_h<float>(const BfArgs&, int, int, int, hipStream_t);
extern template void bf_launch_h<bf16_t>(const BfArgs&, int, int, int, hipStream_t);
// four waves of 128x128, one per SIMD (gemm_bf16_w4.hip): 256x256 tiles only
bool bf_launch_w4(const BfArgs& g, int transA, int transB, hipStream_t s);          // false: no instantiation for this layout

static uint32_t* g_bf_probe = nullptr;
static int g_bf_probe_block = 0, g_bf_probe_mask = 0xff;

template <typename CT>
static void bf_launch(const BfArgs& g0, int big, int transA, int transB, hipStream_t s) {
    BfArgs g = g0;
    g.probe = nullptr; g.probe_block = 0; g.probe_mask = 0;
    int form = opt(OPT_GEMM_BF16_FORM);
    // -1: per launch -- the four-wave 128x128 kernel from K = 2048 up (its shorter k-tile outweighs its ~1.7 us of extra fixed cost per round there:
    // -3 ... -6 % in isolation, profiles/round6_gemm_bf16_wide_ab.log), the eight-wave kernel below.  NOT the default: in the cfg-5 step the choice
    // measured -0.9 % (profiles/round6_cfg5_form_auto_ab.log: beside the other stream's kernels, and with the beta = 1 input-gradient launches on
    // form 4's 2-byte store path)
    if (form < 0) form = (big && !transA && std::is_same<CT, bf16_t>::value && g.K >= 2048 && g.K % KT == 0 && g.splits == 1) ? 4 : 0;
    if (form == 44) { g.probe = g_bf_probe; g.probe_block = g_bf_probe_block; g.probe_mask = g_bf_probe_mask; }
    if constexpr (std::is_same<CT, bf16_t>::value) {
        if (g_bf_probe && big && !transA && transB && form == 0) {          // diagnostics: the forward layout of the 256x256 tile with phase stamps
            g.probe = g_bf_probe; g.probe_block = g_bf_probe_block; g.probe_mask = g_bf_probe_mask;
            hipLaunchKernelGGL((gemm_bf16_kernel<256, 256, true, true, 8, 2, bf16_t, 0, true>), dim3((unsigned)(g.ntiles * g.splits)), dim3(512), 0, s, g);
            return;
        }
    }
    if ((form == 4 || (form >= 41 && form <= 44)) && big && std::is_same<CT, bf16_t>::value && bf_launch_w4(g, transA, transB, s)) return;
    if (form == 3) bf_launch_h<CT>(g, big, transA, transB, s);
    else if (form == 2) bf_launch_form<CT, 2>(g, big, transA, transB, s);
    else if (form == 1) bf_launch_form<CT, 1>(g, big, transA, transB, s);
    else bf_launch_form<CT, 0>(g, big, transA, transB, s);
}

// C = sum_s ws[s] (+ bias) (+ beta*C), fixed summation order -> deterministic; also finishes the per-split row sums of A
template <typename CT>
__global__ __launch_bounds__(256) void bf_splitk_reduce_kernel(const float* __restrict__ ws, CT* __restrict__ C, int64_t ldc,
                                                               const float* __restrict__ bias, int M, int N, int splits, float beta,
                                                               const float* __restrict__ asum_ws, float* __restrict__ asum) {
    if (asum) {
        for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < M; r += (int64_t)gridDim.x * 256) {
            float acc = asum_ws[r];
            for (int s = 1; s < splits; ++s) acc += asum_ws[(int64_t)s * M + r];
            asum[r] = acc;
        }
    }
    const int n4 = (N + 3) >> 2;
    const int64_t total = (int64_t)M * N, groups = (int64_t)M * n4;
    const bool vec = (N & 3) == 0;
    for (int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * 256) {
        const int row = (int)(gidx / n4), col = (int)(gidx % n4) << 2;
        const int64_t i = (int64_t)row * N + col;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const int nv = min(4, N - col);
        if (vec) {
            float4 t = *reinterpret_cast<const float4*>(ws + i);
            for (int s = 1; s < splits; ++s) {
                const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)s * total + i);
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
        } else {
            for (int u = 0; u < nv; ++u) {
                float t = ws[i + u];
                for (int s = 1; s < splits; ++s) t += ws[(int64_t)s * total + i + u];
                a[u] = t;
            }
        }
        CT* cp = C + (int64_t)row * ldc + col;
        for (int u = 0; u < nv; ++u) {
            float v = a[u];
            if (bias) v += bias[col + u];
            if (beta != 0.f) v += beta * Elem<CT>::ld(cp + u);
            Elem<CT>::st(cp + u, v);
        }
    }
}

// ---- generic kernel: any alignment, any size (tiny test configurations, K or leading dimensions that are not multiples of 8) -------------
// 64x64 tile, 256 threads; operands are converted to fp32 on the way into LDS and multiplied on the fp32 matrix instruction: the products of
// bf16 values are exact in fp32 either way, so this is the same arithmetic as the fast path up to the order of the fp32 additions.
template <typename CT>
__global__ __launch_bounds__(256) void gemm_bf16_generic_kernel(const BfArgs g, int transA, int transB) {
    constexpr int BM = 64, BN = 64, BK = 32;
    __shared__ float As[BK][BM + 1], Bs[BK][BN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    float rsum = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        for (int idx = tid; idx < BM * BK; idx += 256) {
            const int k = idx % BK, m = idx / BK;            // consecutive threads along k for a K-contiguous operand ...
            const int km = transA ? idx / BM : k, mm = transA ? idx % BM : m;      // ... along m for a k-major one
            const int gm = m0 + mm, gk = k0 + km;
            float v = 0.f;
            if (gm < g.M && gk < g.K) v = bf2f(transA ? g.A[(int64_t)gk * g.lda + gm] : g.A[(int64_t)gm * g.lda + gk]);
            As[km][mm] = v;
        }
        for (int idx = tid; idx < BN * BK; idx += 256) {
            const int k = idx % BK, n = idx / BK;
            const int kn = transB ? k : idx / BN, nn = transB ? n : idx % BN;
            const int gn = n0 + nn, gk = k0 + kn;
            float v = 0.f;
            if (gn < g.N && gk < g.K) v = bf2f(transB ? g.B[(int64_t)gn * g.ldb + gk] : g.B[(int64_t)gk * g.ldb + gn]);
            Bs[kn][nn] = v;
        }
        __syncthreads();
        if (g.asum && tn == 0 && tid < BM) {          // row sums of op(A) (bias gradient): the first tile column adds up its A tile, k ascending
#pragma unroll 8
            for (int kk = 0; kk < BK; ++kk) rsum += As[kk][tid];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + half][wm0 + l31], Bs[kk + half][wn0 + l31], acc[0][0], 0, 0, 0);
        __syncthreads();
    }
    if (g.asum && tn == 0 && tid < BM && m0 + tid < g.M) g.asum[m0 + tid] = rsum;
    bf_epilogue<1, 1, CT>(g, acc, m0 + wm0, n0 + wn0, l31, half, 0);
}

// out[r][c] = bf16(x[r][c]) (RNE), c < cols; 8 columns per thread where alignment allows.  The one cast of the bf16-resident path: network
// INPUTS that arrive as fp32 (region features) and parameters that are not (yet) inside the optimizer's arenas.
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols, bf16_t* __restrict__ out,
                                                            int64_t ldo, int vec) {
    const int g8 = (cols + 7) >> 3;
    const int64_t total = rows * g8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / g8;
        const int c = (int)(i % g8) << 3;
        const float* p = x + r * ldx + c;
        bf16_t* o = out + r * ldo + c;
        if (vec && c + 8 <= cols) {
            const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
            *reinterpret_cast<uint4*>(o) = make_uint4(f2bf(a.x) | (f2bf(a.y) << 16), f2bf(a.z) | (f2bf(a.w) << 16), f2bf(b.x) | (f2bf(b.y) << 16),
                                                      f2bf(b.z) | (f2bf(b.w) << 16));
        } else {
            for (int u = 0; u < 8 && c + u < cols; ++u) o[u] = (bf16_t)f2bf(p[u]);
        }
    }
}

// ---- launch plan ---------------------------------------------------------------------------------------------------------------------
// With 8x shorter matrix time than the fp32 instruction the operand path is what a workgroup waits for, so the 256x256 tile (half the operand
// bytes per flop, one workgroup per CU) wins wherever it fills the chip about as well as 128x128 does; few output tiles with a long contraction
// (weight gradients) are split along k so that tiles x splits fills one round of the 256 CUs (256x256) or whole rounds of the 512 resident
// 128x128 workgroups.  Fused activations stay unsplit.
struct BfPlan { int big; int splits; };
static BfPlan bf_plan(int M, int N, int K, int epilogue) {
    BfPlan p = {0, 1};
    const int64_t t128 = cdiv(M, 128) * cdiv(N, 128), t256 = cdiv(M, 256) * cdiv(N, 256);
    const int force_tile = opt(OPT_GEMM_TILE), force_sp = opt(OPT_GEMM_SPLITS);
    const bool can_big = M >= 256 && N >= 256;
    if (epilogue == YTVLN_EPI_NONE && K >= 1024) {
        if (can_big && t256 < 200) {
            const int sp = (int)std::min<int64_t>(256 / t256, K / 512);
            if (sp >= 2 && t256 * sp >= 200) { p.big = 1; p.splits = sp; }
        }
        if (!p.big && t128 < 384) {
            const int smax = (int)std::min<int64_t>(64, K / 512);
            double best_eff = (double)t128 / (double)(cdiv(t128, 512) * 512);
            for (int sp = 2; sp <= smax; ++sp) {
                const int kchunk = (int)cdiv(cdiv(K, sp), KT) * KT;
                const int64_t blocks = t128 * cdiv(K, kchunk);
                const double eff = (double)blocks / (double)(cdiv(blocks, 512) * 512);
                if (eff > best_eff + 0.02) { best_eff = eff; p.splits = sp; }
            }
        }
    }
    if (p.splits == 1 && can_big) {
        const double e128 = (double)t128 / (ceil((double)t128 / 512.0) * 512.0), e256 = (double)t256 / (ceil((double)t256 / 256.0) * 256.0);
        p.big = (t256 >= 200 && e256 >= e128 - 0.05) ? 1 : 0;
    }
    if (force_tile == 0) p.big = 0;
    if (force_tile == 4 && can_big) p.big = 1;
    if (force_sp >= 1 && epilogue == YTVLN_EPI_NONE) p.splits = std::max(1, std::min(force_sp, K / KT));
    return p;
}

}  // namespace ytvln

using namespace ytvln;

extern "C" int ytvln_gemm_bf16_probe(uint32_t* buffer, int block, int mask) {
    g_bf_probe = buffer; g_bf_probe_block = block; g_bf_probe_mask = mask;
    return 0;
}

extern "C" int64_t ytvln_gemm_bf16_workspace_elems(int M, int N, int K, int epilogue) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const BfPlan p = bf_plan(M, N, K, epilogue);
    const int splits = std::max(p.splits, std::max(1, opt(OPT_GEMM_SPLITS)));
    return splits > 1 ? (int64_t)splits * M * N + (int64_t)splits * ((M + 3) / 4 * 4) : 0;
}

extern "C" int ytvln_gemm_bf16(const uint16_t* A, int64_t lda, int transA, const uint16_t* B, int64_t ldb, int transB, void* C, int64_t ldc,
                               int c_dtype, const float* bias, uint16_t* aux, int64_t ldaux, int M, int N, int K, int epilogue, float beta,
                               float* workspace, int64_t workspace_elems, int flags, float* a_rowsum, int* rowsum_done, void* stream) {
    if (rowsum_done) *rowsum_done = 0;
    YT_REQUIRE(A && B && C, "gemm_bf16: null operand");
    YT_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_bf16: negative size");
    YT_REQUIRE(c_dtype == YTVLN_DT_F32 || c_dtype == YTVLN_DT_BF16, "gemm_bf16: output type must be YTVLN_DT_F32 or YTVLN_DT_BF16");
    YT_REQUIRE(epilogue >= YTVLN_EPI_NONE && epilogue <= YTVLN_EPI_MUL_DRELU, "gemm_bf16: bad epilogue %d", epilogue);
    YT_REQUIRE(!(epilogue >= YTVLN_EPI_MUL_DGELU) || aux, "gemm_bf16: epilogue %d needs aux", epilogue);
    YT_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "gemm_bf16: leading dimension too small");
    YT_REQUIRE(!a_rowsum || rowsum_done, "gemm_bf16: a_rowsum needs rowsum_done");
    if (M == 0 || N == 0) return 0;
    hipStream_t s = as_stream(stream);
    BfArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
    g.M = M; g.N = N; g.K = K; g.epilogue = epilogue; g.beta = beta;
    g.splits = 1; g.kchunk = (int)cdiv(std::max(K, 1), KT) * KT; g.ws = nullptr; g.asum = nullptr; g.asum_ws = nullptr;
    g.mnA = M; g.mnB = N; g.kvalidA = K; g.kvalidB = K; g.stagger_ticks = 0;
    // wide stores: 1 always where legal, 0 (default) never on the eight-wave kernel (the four-wave form always uses them), -1 where they measured ahead on the eight-wave
    // kernel: outputs that dwarf the contraction (the 30522-wide decoder: -5 %); on the 768 ... 3072-wide projections the LDS round trip costs
    // more than the 2-byte stores it replaces (+1 ... +10 %)
    { const int w = opt(OPT_GEMM_BF16_WIDE); g.wide_stores = w >= 0 ? w : (N >= 8192 && N >= 8 * K); }
    // Fast-path legality: 16-byte aligned operands whose rows start on 16-byte boundaries, and whole 16-byte granules:
    //   contraction-contiguous operand: K % 8 == 0 -- or, for A only, readable ZERO padding behind K up to lda (YTVLN_GEMM_A_ZERO_PADDED: the
    //     30522- and 1601-wide logit gradients as the A operand of the input-gradient GEMM);
    //   k-major operand: its M / N extent % 8 == 0 and >= 8 -- or, for A only, zero padding behind M up to lda (the same gradients as the A operand
    //     of the weight-gradient GEMM); its k rows are predicated one by one, so K is arbitrary.
    const bool apad = (flags & YTVLN_GEMM_A_ZERO_PADDED) != 0;
    const int k8 = (int)cdiv(K, 8) * 8, m8 = (int)cdiv(M, 8) * 8;
    bool ok = K > 0 && lda % 8 == 0 && ldb % 8 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && !opt(OPT_GEMM_GENERIC);
    if (!transA) {
        if (K % 8 != 0) { if (apad && lda >= k8) g.kvalidA = k8; else ok = false; }
    } else {
        if (M < 8) ok = false;
        else if (M % 8 != 0) { if (apad && lda >= m8) g.mnA = m8; else ok = false; }
    }
    if (transB) { if (K % 8 != 0) ok = false; }
    else { if (N % 8 != 0 || N < 8) ok = false; }
    if (!ok) {
        g.tiles_m = (int)cdiv(M, 64); g.tiles_n = (int)cdiv(N, 64); g.ntiles = g.tiles_m * g.tiles_n;
        if (a_rowsum) { g.asum = a_rowsum; *rowsum_done = 1; }
        if (c_dtype == YTVLN_DT_F32) hipLaunchKernelGGL((gemm_bf16_generic_kernel<float>), dim3((unsigned)g.ntiles), dim3(256), 0, s, g, transA, transB);
        else hipLaunchKernelGGL((gemm_bf16_generic_kernel<bf16_t>), dim3((unsigned)g.ntiles), dim3(256), 0, s, g, transA, transB);
        YT_LAUNCH_CHECK("gemm_bf16 (generic)");
        return 0;
    }
    const BfPlan plan = bf_plan(M, N, K, epilogue);
    const int bt = plan.big ? 256 : 128;
    g.tiles_m = (int)cdiv(M, bt); g.tiles_n = (int)cdiv(N, bt); g.ntiles = g.tiles_m * g.tiles_n;
    const bool asum_ok = a_rowsum && transA;
    const int64_t m4r = (M + 3) / 4 * 4;
    if (plan.splits > 1 && workspace && workspace_elems >= (int64_t)plan.splits * M * N + (asum_ok ? (int64_t)plan.splits * m4r : 0)) {
        g.kchunk = (int)cdiv(cdiv(K, plan.splits), KT) * KT;
        g.splits = (int)cdiv(K, g.kchunk);
        g.ws = workspace;
        if (asum_ok) { g.asum = a_rowsum; g.asum_ws = workspace + (int64_t)g.splits * M * N; }
    }
    if (g.splits == 1) {
        g.kchunk = (int)cdiv(K, KT) * KT;
        if (asum_ok) g.asum = a_rowsum;
    }
    if (rowsum_done) *rowsum_done = g.asum != nullptr;
    {   // GEMM_STAGGER = percent of one tile's main-loop time (1.6 us per k-tile on the 256x256 tile): launches of >= 3 rounds only
        const int pct = opt(OPT_GEMM_STAGGER);
        if (pct > 0 && plan.big && (int64_t)g.ntiles * g.splits >= 3 * 256) {
            const int nk = (int)cdiv(std::min(K, g.kchunk), KT);
            g.stagger_ticks = (int)(nk * 1.6 * pct);          // 100 ticks per us, pct / 100 of nk * 1.6 us
        }
    }
    if (c_dtype == YTVLN_DT_F32) bf_launch<float>(g, plan.big, transA, transB, s);
    else bf_launch<bf16_t>(g, plan.big, transA, transB, s);
    if (g.splits > 1) {
        const int64_t total = (int64_t)M * N;
        const dim3 rgrid((unsigned)std::min<int64_t>(cdiv(total, 1024), 2048));
        if (c_dtype == YTVLN_DT_F32)
            hipLaunchKernelGGL((bf_splitk_reduce_kernel<float>), rgrid, dim3(256), 0, s, (const float*)workspace, reinterpret_cast<float*>(C), ldc, bias, M, N,
                               g.splits, beta, (const float*)g.asum_ws, g.asum);
        else
            hipLaunchKernelGGL((bf_splitk_reduce_kernel<bf16_t>), rgrid, dim3(256), 0, s, (const float*)workspace, reinterpret_cast<bf16_t*>(C), ldc, bias, M, N,
                               g.splits, beta, (const float*)g.asum_ws, g.asum);
    }
    YT_LAUNCH_CHECK("gemm_bf16");
    return 0;
}

extern "C" int ytvln_cast_f32_bf16(const float* x, int64_t ldx, int64_t rows, int cols, uint16_t* out, int64_t ldo, void* stream) {
    YT_REQUIRE(x && out && rows >= 0 && cols > 0 && ldx >= cols && ldo >= cols, "cast_f32_bf16: bad argument");
    if (rows == 0) return 0;
    const int vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (ldo % 8 == 0);
    const int64_t total = rows * ((cols + 7) >> 3);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 16384)), dim3(256), 0, as_stream(stream), x, ldx, rows, cols,
                       out, ldo, vec);
    YT_LAUNCH_CHECK("cast_f32_bf16");
    return 0;
}
#endif  // YT_BF16_FORM_TU
#endif  // YT_BF16_SHARED_ONLY
