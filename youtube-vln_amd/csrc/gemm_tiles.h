// Shared device-side pieces of the fp32 GEMM kernels (gemm.hip, gemm_x3.hip, gemm_sk.hip): launch arguments, workgroup -> tile maps,
// the register-staged tile loader of the generic kernel, the LDS-DMA tile images (DmaTile) and the fused epilogues.
#pragma once
#include "common.h"
#include <algorithm>
#include <vector>
#include <type_traits>
#include <utility>
#include <stdlib.h>

namespace ytvln {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias; float* aux;
    int64_t lda, ldb, ldc, ldaux;
    int M, N, K;
    int epilogue;
    float beta;
    int vecA, vecB;   // 16-byte aligned vector loads legal for the operand
    int tiles_m, tiles_n, ntiles;
    int splits, kchunk;   // split-K: blockIdx.y owns k in [y*kchunk, (y+1)*kchunk); partial tiles go to ws[y][M][N]
    float* ws;
    int fast;             // LDS-DMA main loop legal (K % 32 == 0, aligned operands, M/N-contiguous extents % 4 == 0)
    int Kloop;            // contraction length the fast kernel iterates over (K rounded up to 32 when A's K tail is zero-padded)
    int mnA, mnB;         // clamp extents of the operands in their M / N dimension (rounded up to 4 inside padding)
    int ktail;            // 1: the last k-tile reaches past K -> M/N-contiguous operands clamp their k rows to K-1
    int x3;               // 1: fp32 operands split into three bf16 terms in registers, six bf16 MFMAs per product (YTVLN_GEMM_SPLIT_BF16X3)
    float* asum;          // optional: asum[m] = sum_k op(A)[m, k] (bias gradient riding on the weight-gradient GEMM); M-contiguous A, LDS-DMA path only
    float* asum_ws;       // split-K: per-split partial row sums [splits][M], reduced in a fixed order by splitk_reduce_kernel
    int split_map;        // 1: split-K workgroups are laid out split-major per XCD (see decode_tile)
};

constexpr int BK = 32;

// Workgroup id -> (tile row, tile column, split).  MI355X dispatches workgroup b to XCD b % 8 and every XCD has its own L2:
//  * split-K launches make the SPLIT the fastest-varying index, so (for 8 splits) each XCD streams one disjoint K-slab of
//    both operands for all output tiles -- instead of every XCD re-reading the whole of B;
//  * otherwise each XCD gets a contiguous range of tiles (xcd_remap), visited in groups of 8 tile rows x all columns taken
//    column-by-column ("grouped" order), so the ~64 tiles resident on an XCD form an 8 x 8 patch that shares 8 A panels and
//    8 B panels instead of 2-3 rows x 24 columns.
struct TileCoord { int m, n, split; };
// linear tile index (already XCD-local) -> tile row / column in "grouped" order
__device__ __forceinline__ TileCoord tile_coord(int t, int tiles_m, int tiles_n) {
    TileCoord c;
    c.split = 0;
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int g = t / per_group, first_m = g * GROUP_M;
    const int rows = min(tiles_m - first_m, GROUP_M);
    const int r = t - g * per_group;
    c.m = first_m + r % rows;
    c.n = r / rows;
    return c;
}
__device__ __forceinline__ TileCoord decode_tile(int bid, int tiles_m, int tiles_n, int splits, int split_major = 1) {
    // Workgroups that read the same operand panels must meet in one XCD's L2 at about the same time.  Without split-K an XCD gets a
    // contiguous run of the (group-major) tile order; with split-K it gets a contiguous run of the split-major (split, tile) order, i.e.
    // neighbouring tiles of ONE k range (they share its A / B panels) -- not, as before round 2, all splits of one tile scattered over the
    // XCDs by bid % 8 (L2 hit rate 0.31 -> see profiles/round2_gemm_traffic.json; YTVLN_GEMM_SPLIT_MAP=0 restores that order).
    int t, split;
    const int ntiles = tiles_m * tiles_n;
    if (splits > 1 && split_major) { const int id = xcd_remap(bid, ntiles * splits); split = id / ntiles; t = id - split * ntiles; }
    else if (splits > 1) { split = bid % splits; t = bid / splits; }
    else { split = 0; t = xcd_remap(bid, ntiles); }
    TileCoord c = tile_coord(t, tiles_m, tiles_n);
    c.split = split;
    return c;
}

template <int BMN, bool KC>
struct TileLoader {
    // number of float4 per thread for a BMN x 32 tile with 256 threads
    static constexpr int NV = BMN * BK / 4 / 256;
    static constexpr int LD = KC ? BMN + 1 : BMN;

    // global -> registers. mn0: first row/col of the tile in the M/N dimension, k0: first k.
    __device__ static __forceinline__ void load(float4 (&r)[NV], const float* __restrict__ P, int64_t ld, int MN, int K,
                                                int mn0, int k0, int vec, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + 256 * j;
            int mn, k;
            if (KC) { mn = idx >> 3; k = (idx & 7) << 2; }                       // 8 float4 along k per row
            else { k = idx / (BMN / 4); mn = (idx % (BMN / 4)) << 2; }           // BMN/4 float4 along mn per k-row
            const int gmn = mn0 + mn, gk = k0 + k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                if (gmn < MN && gk < K) {
                    const float* p = P + (int64_t)gmn * ld + gk;
                    if (vec && gk + 3 < K) v = *reinterpret_cast<const float4*>(p);
                    else {
                        v.x = p[0];
                        if (gk + 1 < K) v.y = p[1];
                        if (gk + 2 < K) v.z = p[2];
                        if (gk + 3 < K) v.w = p[3];
                    }
                }
            } else {
                if (gk < K && gmn < MN) {
                    const float* p = P + (int64_t)gk * ld + gmn;
                    if (vec && gmn + 3 < MN) v = *reinterpret_cast<const float4*>(p);
                    else {
                        v.x = p[0];
                        if (gmn + 1 < MN) v.y = p[1];
                        if (gmn + 2 < MN) v.z = p[2];
                        if (gmn + 3 < MN) v.w = p[3];
                    }
                }
            }
            r[j] = v;
        }
    }
    // registers -> LDS (k-major image S[k][mn])
    __device__ static __forceinline__ void store(const float4 (&r)[NV], float* __restrict__ S, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + 256 * j;
            if (KC) {
                const int mn = idx >> 3, k = (idx & 7) << 2;
                S[(k + 0) * LD + mn] = r[j].x;
                S[(k + 1) * LD + mn] = r[j].y;
                S[(k + 2) * LD + mn] = r[j].z;
                S[(k + 3) * LD + mn] = r[j].w;
            } else {
                const int k = idx / (BMN / 4), mn = (idx % (BMN / 4)) << 2;
                *reinterpret_cast<float4*>(S + k * LD + mn) = r[j];
            }
        }
    }
};

// Epilogue shared by both main loops: lane owns column l31 of each 32x32 tile, rows (r&3) + 8*(r>>2) + 4*half.
// The epilogue kind and the "tile lies inside the matrix" test are resolved ONCE per wave (uniform branches around fully
// specialised store loops): the per-element switch / bounds tests of a generic loop cost more than the stores themselves.
template <int TM, int TN, int EPI, bool INTERIOR>
__device__ __forceinline__ void epilogue_body(const GemmArgs& g, f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int half) {
    const bool has_beta = g.beta != 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + 32 * j + l31;
        if (!INTERIOR && col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = row0 + 32 * i + 4 * half;
            float* cp0 = g.C + (int64_t)rbase * g.ldc + col;
            float* xp0 = (EPI != YTVLN_EPI_NONE && EPI != YTVLN_EPI_RELU) ? g.aux + (int64_t)rbase * g.ldaux + col : nullptr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {          // 4 rows (dr = 8q + 0..3) at a time: loads in flight together, few live registers
                float ax[4], old[4];
                if (EPI == YTVLN_EPI_MUL_DGELU || EPI == YTVLN_EPI_MUL_DRELU) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ax[u] = (INTERIOR || rbase + 8 * q + u < g.M) ? xp0[(int64_t)(8 * q + u) * g.ldaux] : 0.f;
                }
                if (has_beta) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) old[u] = (INTERIOR || rbase + 8 * q + u < g.M) ? cp0[(int64_t)(8 * q + u) * g.ldc] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int dr = 8 * q + u;
                    if (!INTERIOR && rbase + dr >= g.M) continue;
                    float v = acc[i][j][4 * q + u] + bv;
                    if (EPI == YTVLN_EPI_GELU) {
                        if (g.aux) xp0[(int64_t)dr * g.ldaux] = v;
                        v = gelu_erf(v);
                    } else if (EPI == YTVLN_EPI_RELU) {
                        v = fmaxf(v, 0.f);
                    } else if (EPI == YTVLN_EPI_MUL_DGELU) {
                        v *= dgelu_erf(ax[u]);
                    } else if (EPI == YTVLN_EPI_MUL_DRELU) {
                        v = ax[u] > 0.f ? v : 0.f;
                    }
                    if (has_beta) v += g.beta * old[u];
                    cp0[(int64_t)dr * g.ldc] = v;
                }
            }
        }
    }
}

// Interior tiles (every row and column of the wave's sub-tile inside the matrix): the same arithmetic with the addressing rebuilt for the
// register file.  row0 / col0 are wave-uniform, so the address of every store is (scalar base of its row) + ONE per-lane byte offset that is the
// same for the whole epilogue (4 * half rows down, l31 columns right): `global_store_dword voff, data, s[base]`.  The generic body above keeps a
// 64-bit pointer per row group alive next to the 128 accumulators; hipcc then spills a handful of them and -- scratch reloads and global stores
// share the in-order vmcnt counter -- every reload waits for ALL stores issued before it: ~30 store round trips per 256x256 tile (round 5
// timeline: 29 us for the epilogue of a 256x256 tile against 4 us for a 256x128 one; this was the launch-per-tile kernel's 25 us "fixed cost").
// Interior tiles (every row and column of the wave's sub-tile inside the matrix): the arithmetic of epilogue_body with the addressing rebuilt for
// the register file.  row0 / col0 are wave-uniform, so every access is (scalar base of its row) + ONE per-lane byte offset that is the same for
// the whole epilogue (4 * half rows down, l31 columns right).  The generic body keeps 64-bit per-lane pointers alive next to the 128 accumulators
// of a 256x256 tile; hipcc spilled a handful of them and -- scratch reloads and global stores share the in-order vmcnt counter -- every reload
// waited for ALL stores issued before it: ~30 store round trips per tile (round-5 timeline: 29 us for the epilogue of a 256x256 tile against 4 us
// for a 256x128 one; this was the launch-per-tile kernel's 25 us of "fixed cost").  Epilogues that READ a matrix (beta, x GELU', x ReLU') are
// software-pipelined: the loads of row group g + 1 are issued before the stores of group g, and the wait counts the operations behind them.
template <int TM, int TN, int EPI, bool BETA>
__device__ __forceinline__ void epilogue_interior(const GemmArgs& g, f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int half) {
    constexpr bool AUX_IN = EPI == YTVLN_EPI_MUL_DGELU || EPI == YTVLN_EPI_MUL_DRELU;
    constexpr int NL = (AUX_IN ? 4 : 0) + (BETA ? 4 : 0);         // loads per row group
    constexpr int NST = 4;                                        // stores per row group (8 when GELU also writes its pre-activation: the wait below stays conservative)
    constexpr int G = TN * TM * 4;
    const uint32_t vc = (uint32_t)(((int64_t)(4 * half) * g.ldc + l31) * 4);          // per-lane byte offsets (4 rows + 32 columns: < 2^32)
    const uint32_t vx = (uint32_t)(((int64_t)(4 * half) * g.ldaux + l31) * 4);
    const char* const cb = scalar_ptr(reinterpret_cast<const char*>(g.C + (int64_t)row0 * g.ldc + col0));
    const char* const xb = (EPI != YTVLN_EPI_NONE && EPI != YTVLN_EPI_RELU && g.aux) ? scalar_ptr(reinterpret_cast<const char*>(g.aux + (int64_t)row0 * g.ldaux + col0)) : nullptr;
    const int64_t cs = g.ldc * 4, xs = g.ldaux * 4;               // row strides in bytes
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = g.bias ? g.bias[col0 + 32 * j + l31] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bv[j]));          // the bias loads are the compiler's: its waits land here, before the hand-counted part
    float ax[2][4] = {}, old[2][4] = {};
    // (each group's row base goes through an opaque asm: left alone, hipcc works out all 128 row bases of the epilogue at once, needs ~250 scalar
    //  registers for them and spills scalars into VGPR lanes -- inside a persistent kernel even in the main loop)
    auto row_c = [&](int gi) { const int j = gi / (TM * 4), i = (gi / 4) % TM, q = gi % 4; const char* p = cb + (int64_t)(32 * i + 8 * q) * cs + 128 * j; asm volatile("" : "+s"(p)); return p; };
    auto row_x = [&](int gi) { const int j = gi / (TM * 4), i = (gi / 4) % TM, q = gi % 4; const char* p = xb + (int64_t)(32 * i + 8 * q) * xs + 128 * j; asm volatile("" : "+s"(p)); return p; };
    auto loads = [&](auto gc) {
        constexpr int gi = decltype(gc)::value;
        if constexpr (AUX_IN) {
            const char* const px = row_x(gi);
#pragma unroll
            for (int u = 0; u < 4; ++u) epi_load(ax[gi & 1][u], vx, px + u * xs);
        }
        if constexpr (BETA) {
            const char* const pc = row_c(gi);
#pragma unroll
            for (int u = 0; u < 4; ++u) epi_load(old[gi & 1][u], vc, pc + u * cs);
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
                if (px) epi_store(vx, v, px + u * xs);
                v = gelu_erf(v);
            } else if (EPI == YTVLN_EPI_RELU) {
                v = fmaxf(v, 0.f);
            } else if (EPI == YTVLN_EPI_MUL_DGELU) {
                v *= dgelu_erf(ax[gi & 1][u]);
            } else if (EPI == YTVLN_EPI_MUL_DRELU) {
                v = ax[gi & 1][u] > 0.f ? v : 0.f;
            }
            if (BETA) v += g.beta * old[gi & 1][u];
            epi_store(vc, v, pc + u * cs);
        }
    });
}

// HAND = false: interior tiles that READ a matrix (beta, x GELU', x ReLU') take epilogue_body's compiler-counted loads instead of the
// hand-counted ones of epilogue_interior.  For kernels that are allowed a scratch segment (_build.py SCRATCH_OK): a register the allocator
// spills or copies between a hand-issued load and its hand-counted wait is read before the data lands.
template <int TM, int TN, bool HAND = true>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int half, int split) {
    const bool interior = row0 + 32 * TM <= g.M && col0 + 32 * TN <= g.N;
    if (g.splits > 1) {      // raw partial sums; bias / beta are applied by splitk_reduce_kernel in a fixed order
        float* w = g.ws + (int64_t)split * g.M * g.N;
        if (interior) {          // scalar row bases + one per-lane offset (see epilogue_interior)
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
#define YT_EPI(E)                                                                        \
    case E:                                                                              \
        if (interior) {                                                                  \
            constexpr bool READS = E == YTVLN_EPI_MUL_DGELU || E == YTVLN_EPI_MUL_DRELU;  \
            if (!HAND && (READS || g.beta != 0.f)) epilogue_body<TM, TN, E, true>(g, acc, row0, col0, l31, half); \
            else if (g.beta != 0.f) epilogue_interior<TM, TN, E, true>(g, acc, row0, col0, l31, half); else epilogue_interior<TM, TN, E, false>(g, acc, row0, col0, l31, half); } \
        else epilogue_body<TM, TN, E, false>(g, acc, row0, col0, l31, half);             \
        break;
    switch (g.epilogue) {
        YT_EPI(YTVLN_EPI_GELU)
        YT_EPI(YTVLN_EPI_RELU)
        YT_EPI(YTVLN_EPI_MUL_DGELU)
        YT_EPI(YTVLN_EPI_MUL_DRELU)
        default:
            if (interior) {
                if (!HAND && g.beta != 0.f) epilogue_body<TM, TN, YTVLN_EPI_NONE, true>(g, acc, row0, col0, l31, half);
                else if (g.beta != 0.f) epilogue_interior<TM, TN, YTVLN_EPI_NONE, true>(g, acc, row0, col0, l31, half); else epilogue_interior<TM, TN, YTVLN_EPI_NONE, false>(g, acc, row0, col0, l31, half); }
            else epilogue_body<TM, TN, YTVLN_EPI_NONE, false>(g, acc, row0, col0, l31, half);
            break;
    }
#undef YT_EPI
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int BMN, bool KC, int NW, int KB>
struct DmaTile {
    static constexpr int PIECES = BMN * KB / 256;   // 1 KiB pieces per k-tile
    static constexpr int NI = (PIECES + NW - 1) / NW;   // pieces per wave (wave w owns pieces w*NI .. w*NI+NI-1)
    static constexpr bool RAGGED = PIECES % NW != 0;    // 224-row tile: 28 pieces over 8 waves -- the last wave(s) own fewer; callers test `own`
    __device__ static __forceinline__ bool own(int wave, int i) { return !RAGGED || wave * NI + i < PIECES; }
    static constexpr int GR = KB / 4;               // 16-byte granules per row of a K-contiguous image (8 or 4)
    static constexpr int RP = 256 / KB;             // rows per 1 KiB piece of a K-contiguous image (8 or 16)
    static_assert(NI >= 1 && (KB == 64 || KB == 32 || KB == 16), "unsupported DMA tile");
    // XOR swizzle of a K-contiguous image: chosen so that the 16 rows of each ds_read_b128 lane group
    // ({0-3,12-15,20-27} / {4-11,16-19,28-31}) fall on 16 distinct 16-byte slots of the 256-byte bank row.
    __device__ static __forceinline__ int swz(int m) { return KB == 64 ? (m & 15) : KB == 32 ? ((m >> 1) & 7) : ((m >> 2) & 3); }
    // per-lane source pointer of piece i for the tile starting at k0 (advanced by the caller)
    // MN: extent used for clamping rows (K-contiguous operand) / 16-byte column granules (M/N-contiguous operand);
    // kmax: last valid k row of an M/N-contiguous operand (rows past it are clamped: they meet zero padding of the other
    // operand's K tail, see YTVLN_GEMM_A_ZERO_PADDED).
    __device__ static __forceinline__ const float* src(const float* P, int64_t ld, int MN, int mn0, int k0, int wave, int lane, int i,
                                                       int kmax) {
        const int c = wave * NI + i;
        if (KC) {
            const int m = c * RP + lane / GR;
            const int g = (lane % GR) ^ swz(m);
            const int row = min(mn0 + m, MN - 1);
            return P + (int64_t)row * ld + k0 + 4 * g;
        } else {
            const int per_row = BMN / 4;                       // float4 per k-row
            const int k = (c * 64 + lane) / per_row, mn = ((c * 64 + lane) % per_row) * 4;
            const int col = min(mn0 + mn, MN - 4);
            return P + (int64_t)min(k0 + k, kmax) * ld + col;
        }
    }
    __device__ static __forceinline__ int64_t step(int64_t ld) { return KC ? KB : KB * ld; }   // floats per k-tile
    // operand values for k-group sg (k = (KB/2)*half + 4*sg + 0..3) of the wave sub-tile starting at row/col w0 + 32*i
    __device__ static __forceinline__ float4 frag(const float* __restrict__ S, int w0, int i, int l31, int half, int sg) {
        if (KC) {
            const int row = w0 + 32 * i + l31;
            return *reinterpret_cast<const float4*>(S + row * KB + 4 * ((half * (GR / 2) + sg) ^ swz(row)));
        } else {
            const float* p = S + ((KB / 2) * half + 4 * sg) * BMN + w0 + 32 * i + l31;
            return make_float4(p[0], p[BMN], p[2 * BMN], p[3 * BMN]);
        }
    }
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// ---- one-wave-per-SIMD kernels (gemm_sw.hip): false when no instantiation exists for the tile -----------------------------------------
bool launch_sw(int bm, int bn, const GemmArgs& g, int transA, int transB, unsigned grid, hipStream_t s);

// ---- persistent / stream-K kernel (gemm_sk.hip) -------------------------------------------------------------------------------------
struct SkPlan { int use; int tile; int dp; int G; int ngroups; double cost; };       // tile: 3 = 256x128, 4 = 256x256; cost: model estimate, us
SkPlan plan_sk(int M, int N, int Kloop, int transA, int epilogue, bool fast, bool x3);
int sk_num_cus();
int64_t sk_workspace_elems(const SkPlan& p);                                          // floats of partial-tile scratch (0 for the whole-tile form)
void sk_launch(GemmArgs& g, const SkPlan& p, int transB, float* partials, unsigned* ctl, hipStream_t s);

}  // namespace ytvln
