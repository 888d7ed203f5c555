// Fused multi-head attention (forward + backward) on the CDNA4 fp32 matrix cores.
//
// One code path serves BertSelfAttention, BertImageSelfAttention and both directions of BertBiAttention
// (vilbert/vilbert.py:284-311, 413-440, 552-618): queries and keys/values may come from different streams (Tq != Tk) and
// are read in place from the packed projection outputs through (pointer, leading dimension) pairs.
//
// Register-resident flash attention, designed around the 32x32x2 f32 MFMA fragment layout so that no score ever leaves
// the register file and no cross-lane shuffle is needed beyond one half-wave exchange:
//   * a wave owns 32 queries; it computes the TRANSPOSED score tile  S^T[key][query] = K . Q^T  so that the accumulator
//     layout (column = lane&31 = query, 16 keys down the registers, the other 16 keys in the partner half-wave) makes the
//     softmax reduction over keys a per-lane loop + one __shfl_xor(32);
//   * the probabilities are consumed straight from those registers as the B operand of  O^T[dcol][query] += V^T . P^T,
//     because "reg r of lane l" is exactly the (k = key, column = query) element the MFMA wants when the contraction
//     index is visited in the permuted order key(r, half) = (r&3) + 8*(r>>2) + 4*half  (a sum is order-free);
//   * the contraction over the head dimension is split between half-waves (half 0: dk in [0, DP/2), half 1: the rest) so
//     each lane fetches its K operand with ds_read_b128 from an XOR-swizzled row (<= 2-way conflict, section LDS of the guide);
//   * O^T keeps the query on the lane, so the online-softmax rescale and the final 1/l are per-lane scalars.
// K/V tiles of 32 keys are streamed through a two-stage LDS ring by LDS-DMA, shared by the NW waves (32*NW queries) of a workgroup.
//
// Backward = recompute-based flash backward split in two kernels with the same fragment tricks:
//   dq kernel  (workgroup owns queries, loops over key tiles):  S^T, dP^T = V . dO^T, dS^T -> dQ^T += K^T . dS^T
//   dkv kernel (workgroup owns keys,    loops over query tiles): S, dP = dO . V^T, -> dV^T += dO^T . P~, dK^T += Q^T . dS
// The head dimension d may be any multiple of 4 up to 128; it is zero-padded to DP in {32, 64, 128}.
#include "common.h"
#include <algorithm>

namespace ytvln {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const float* q; const float* k; const float* v; const float* mask;
    const float* ctx; const float* dctx; const float* lse; const float* delta;
    float* out; float* lse_out; float* dq; float* dk; float* dv;
    int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int N, heads, Tq, Tk, d;
    float scale, p_drop;
    const int64_t* rng; int64_t site;
    int bf16;          // 1: bf16 MFMA operands (ytvln_attn_*_bf16 entry points)
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define RESCALE_THR 12.0f    // e^12 ~ 1.6e5: far inside fp32 range even summed over thousands of keys

__device__ __forceinline__ int krow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// scaled + masked score with the reference's two roundings (scores / sqrt(d), then + mask; vilbert.py:295-297): the
// additive mask is -10000, so an fma here would change fully-masked rows at the 1e-3 level.
__device__ __forceinline__ float score(float s, float scale, float mask) { return __fadd_rn(__fmul_rn(s, scale), mask); }

// ---- LDS tile streaming ------------------------------------------------------------------------------------------------
// A 32 x DP tile lives in LDS as S[row][DP] with the 16-byte granule g of row r stored at position g ^ (r & 7).  Tiles are
// fed by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a two-stage ring, so the next tile is in
// flight under the current tile's MFMAs; the DMA writes lane-linear 1 KiB pieces, so the swizzle is applied to the per-lane
// SOURCE address.  Rows past the end / columns past d are CLAMPED to valid data (finite garbage): such keys carry a -inf
// mask (p = 0), such queries a +inf lse (p = 0), and garbage columns only reach accumulator columns that are never stored.
// Reads: 4 consecutive dk of one row = one ds_read_b128 (<= 2-way conflict); 32 consecutive columns of one row = ds_read_b32.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int DP>
struct Tile {
    static constexpr int PIECES = DP / 8;    // 1 KiB pieces per 32-row tile
    __device__ static __forceinline__ void issue(float* S, const float* __restrict__ base, int64_t ld, int64_t row_base, int row0,
                                                 int nrows, int col0, int d, int wave, int nw, int lane) {
        for (int p = wave; p < PIECES; p += nw) {          // wave-uniform
            const int off = p * 256 + 4 * lane;
            const int row = off / DP, pos = (off % DP) >> 2;
            const int g = pos ^ (row & 7);
            const int grow = min(row0 + row, nrows - 1), gcol = min(4 * g, d - 4);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (row_base + grow) * ld + col0 + gcol), (lds_ptr_t)(S + p * 256), 16, 0, 0);
        }
    }
    __device__ static __forceinline__ float4 row4(const float* S, int row, int gq) {
        return *reinterpret_cast<const float4*>(S + row * DP + 4 * (gq ^ (row & 7)));
    }
    __device__ static __forceinline__ float elem(const float* S, int row, int col) {
        return S[row * DP + 4 * ((col >> 2) ^ (row & 7)) + (col & 3)];
    }
};

// Per-lane LDS offsets, computed once per kernel: the swizzle only touches the low 3 granule bits, so every fragment read
// becomes (one of a few lane-constant bases) + (compile-time immediate).
struct LaneOff {
    int rows[8];   // rows[u]  : float offset of granule (u ^ (l31&7)) of row l31                      -> mma_rows
    int cols[4];   // cols[u]  : float offset of column l31 in a row whose (row & 7) == ((u + 4*half) & 7), incl. 4*half rows -> mma_cols
};
template <int DP>
__device__ __forceinline__ LaneOff make_lane_off(int l31, int half) {
    LaneOff o;
#pragma unroll
    for (int u = 0; u < 8; ++u) o.rows[u] = l31 * DP + 4 * ((half * (DP / 8) + u) ^ (l31 & 7));   // (for DP = 32 the half bit is swizzled too)
#pragma unroll
    for (int u = 0; u < 4; ++u) o.cols[u] = (u + 4 * half) * DP + 4 * ((l31 >> 2) ^ ((u + 4 * half) & 7)) + (l31 & 3);
    return o;
}

// LDS reads in the tile loops go through __restrict__ parameters: a ds_read without alias information makes hipcc emit
// s_waitcnt vmcnt(0) in front of it whenever an LDS-DMA is in flight (it might alias the DMA's destination), which drains the
// prefetch of the next tile in the middle of the current one.
__device__ __forceinline__ float4 lds4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }

// per-lane operand registers: X[row0 + (lane&31)][half*(DP/2) + s], s = 0..DP/2-1 (zero past nrows / past d)
template <int DP>
__device__ __forceinline__ void load_rowfrag(float (&R)[DP / 2], const float* __restrict__ base, int64_t ld, int64_t row_base,
                                             int row, int nrows, int col0, int d, int half) {
#pragma unroll
    for (int s4 = 0; s4 < DP / 2; s4 += 4) {
        const int col = half * (DP / 2) + s4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nrows && col < d) v = *reinterpret_cast<const float4*>(base + (row_base + row) * ld + col0 + col);
        R[s4] = v.x; R[s4 + 1] = v.y; R[s4 + 2] = v.z; R[s4 + 3] = v.w;
    }
}

// bf16-operand variants (opt-in "bf16 MFMA path", BASELINE config 5): tiles and register fragments stay fp32; eight consecutive
// contraction values of a lane are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on the way into ONE v_mfma_f32_32x32x16_bf16 where the
// fp32 path issues eight 32x32x2 instructions.  Accumulators, softmax, lse / delta and every stored tensor remain fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 pack8(float a, float b, float c, float d, float e, float f, float g, float h) {
    bf16x8 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    v[4] = (__bf16)e; v[5] = (__bf16)f; v[6] = (__bf16)g; v[7] = (__bf16)h;
    return v;
}
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// acc (32x32) = Xs-tile (rows = lane&31 of the LDS tile, contraction split by half) . Rfrag^T
template <int DP, bool BF = false>
__device__ __forceinline__ f32x16 mma_rows(const float* __restrict__ Xs, const float (&R)[DP / 2], const LaneOff& lo) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (BF) {
#pragma unroll
        for (int s8 = 0; s8 < DP / 2; s8 += 8) {       // this half-wave's contraction values s8 .. s8+7 (two 16-byte granules of the row)
            const int g0 = s8 >> 2, g1 = g0 + 1;
            const float4 x0 = *reinterpret_cast<const float4*>(Xs + lo.rows[g0 & 7] + (g0 & ~7) * 4);
            const float4 x1 = *reinterpret_cast<const float4*>(Xs + lo.rows[g1 & 7] + (g1 & ~7) * 4);
            acc = MFMA_BF(pack8(x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w),
                          pack8(R[s8], R[s8 + 1], R[s8 + 2], R[s8 + 3], R[s8 + 4], R[s8 + 5], R[s8 + 6], R[s8 + 7]), acc);
        }
        return acc;
    }
#pragma unroll
    for (int s4 = 0; s4 < DP / 2; s4 += 4) {
        // granule index within the half = s4/4; its low 3 bits are swizzled (lane-constant table), the rest is an immediate
        const float4 x = *reinterpret_cast<const float4*>(Xs + lo.rows[(s4 >> 2) & 7] + ((s4 >> 2) & ~7) * 4);
        acc = MFMA(x.x, R[s4], acc);
        acc = MFMA(x.y, R[s4 + 1], acc);
        acc = MFMA(x.z, R[s4 + 2], acc);
        acc = MFMA(x.w, R[s4 + 3], acc);
    }
    return acc;
}

// acc[c] (dcol x lane-col) += Xs^T (rows = dcol, contraction over the 32 tile rows in krow order) . P (own registers)
template <int DP, bool BF = false>
__device__ __forceinline__ void mma_cols(f32x16 (&acc)[DP / 32], const float* __restrict__ Xs, const float (&P)[16],
                                         const LaneOff& lo, int d) {
    if constexpr (BF) {
#pragma unroll
        for (int c = 0; c < DP / 32; ++c) {
            if (c * 32 < d) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {      // tile rows krow(r, half), r = 8s .. 8s+7: the same order in the P registers
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = 8 * s + j;
                        x[j] = Xs[lo.cols[r & 3] + 8 * (r >> 2) * DP + c * 32];
                    }
                    acc[c] = MFMA_BF(pack8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]),
                                     pack8(P[8 * s], P[8 * s + 1], P[8 * s + 2], P[8 * s + 3], P[8 * s + 4], P[8 * s + 5], P[8 * s + 6],
                                           P[8 * s + 7]), acc[c]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < DP / 32; ++c) {
        if (c * 32 < d) {
#pragma unroll
            for (int r = 0; r < 16; ++r)     // row = (r&3) + 8*(r>>2) + 4*half: (row&7) only depends on (r&3, half)
                acc[c] = MFMA(Xs[lo.cols[r & 3] + 8 * (r >> 2) * DP + c * 32], P[r], acc[c]);
        }
    }
}

// O^T-style accumulators -> global rows (row = this lane's query/key), scaled
template <int DP>
__device__ __forceinline__ void store_cols(const f32x16 (&acc)[DP / 32], float* __restrict__ base, int64_t ld, int64_t grow,
                                           int col0, int d, int half, float mul) {
#pragma unroll
    for (int c = 0; c < DP / 32; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = c * 32 + 8 * g + 4 * half;
            if (col < d)
                *reinterpret_cast<float4*>(base + grow * ld + col0 + col) =
                    make_float4(acc[c][4 * g] * mul, acc[c][4 * g + 1] * mul, acc[c][4 * g + 2] * mul, acc[c][4 * g + 3] * mul);
        }
}

#define TILE_WAIT_AND_SYNC()                                   \
    do {                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_s_barrier();                          \
    } while (0)

// ------------------------------------------------------------------------------------------------------------------
template <int DP, bool DROP, bool BF>
__device__ __forceinline__ void attn_fwd_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP;                 // floats per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [stage0: K V][stage1: K V][mask row, -inf past Tk]
    float* Mrow = smem + 4 * TS;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int qi = (bx * nw + wave) * 32 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const LaneOff lo = make_lane_off<DP>(l31, half);

    auto issue = [&](int t) {
        float* st = smem + (t & 1) * 2 * TS;
        Tile<DP>::issue(st, a.k, a.ldk, (int64_t)n * a.Tk, t * 32, a.Tk, col0, a.d, wave, nw, lane);
        Tile<DP>::issue(st + TS, a.v, a.ldv, (int64_t)n * a.Tk, t * 32, a.Tk, col0, a.d, wave, nw, lane);
    };
    issue(0);                                   // first K/V tile travels while the Q fragment and the mask row are fetched

    float Qr[DP / 2];
    load_rowfrag<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, a.d, half);
    for (int j = tid; j < ntiles * 32; j += nthr) Mrow[j] = j < a.Tk ? (a.mask ? a.mask[(int64_t)n * a.Tk + j] : 0.f) : -INFINITY;

    f32x16 O[DP / 32];
#pragma unroll
    for (int c = 0; c < DP / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    const uint32_t dlo = (uint32_t)((((int64_t)n * a.heads + h) * a.Tq + qi));      // score row id; element = (row id, key)
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    for (int t = 0; t < ntiles; ++t) {
        TILE_WAIT_AND_SYNC();                      // tile t landed for everyone; everyone finished reading the other stage
        if (t + 1 < ntiles) issue(t + 1);
        const float* Ks = smem + (t & 1) * 2 * TS;
        const float* Vs = Ks + TS;
        const int j0 = t * 32;

        const f32x16 S = mma_rows<DP, BF>(Ks, Qr, lo);
        float P[16];
        float mt = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mk = lds4(Mrow + j0 + 8 * g + 4 * half);
            P[4 * g] = score(S[4 * g], a.scale, mk.x); P[4 * g + 1] = score(S[4 * g + 1], a.scale, mk.y);
            P[4 * g + 2] = score(S[4 * g + 2], a.scale, mk.z); P[4 * g + 3] = score(S[4 * g + 3], a.scale, mk.w);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, P[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        // Online softmax with a lazily moved reference point: the running reference m only moves (and O, l are only
        // rescaled -- 64 accumulator registers through the VALU) when some row's tile maximum exceeds it by more than
        // RESCALE_THR.  Mathematically identical (any reference cancels in O / l); exp arguments stay <= RESCALE_THR.
        if (__any(mt > m + RESCALE_THR)) {
            const float mn = fmaxf(m, mt);
            const float alpha = __expf(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int c = 0; c < DP / 32; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[c][r] *= alpha;
        }
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { P[r] = __expf(P[r] - m); ps += P[r]; }
        ps += __shfl_xor(ps, 32, 64);
        l += ps;
        if (DROP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t bits = attn_drop_hash((uint32_t)(j0 + krow(r, half)), dlo, key);
                P[r] = bits >= thr ? P[r] * ik : 0.f;
            }
        }
        mma_cols<DP, BF>(O, Vs, P, lo, a.d);
    }
    if (qvalid) {
        store_cols<DP>(O, a.out, a.ldo, (int64_t)n * a.Tq + qi, col0, a.d, half, 1.0f / l);
        if (half == 0) a.lse_out[((int64_t)n * a.heads + h) * a.Tq + qi] = m + logf(l);   // lse is reference-independent
    }
}

// delta[n,h,q] = sum_c dctx[n,q,h*d+c] * ctx[n,q,h*d+c]
// LG lanes (a power of two >= d/4) share one (row, head) segment: every lane moves one 16-byte piece of ctx and dctx, so a wave
// sweeps 1 KiB of each row contiguously; the segment sum is a log2(LG)-step shuffle reduction (fixed order -> deterministic).
template <int LG>
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ ctx, const float* __restrict__ dctx, int64_t ldo,
                                                         float* __restrict__ delta, int N, int heads, int Tq, int d) {
    const int64_t total = (int64_t)N * Tq * heads;
    const int sub = threadIdx.x % LG;
    const int d4 = d >> 2;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LG; i < total; i += (int64_t)gridDim.x * (256 / LG)) {
        const int h = (int)(i % heads);
        const int64_t row = i / heads;   // n*Tq + q
        float acc = 0.f;
        if (sub < d4) {
            const float4 x = reinterpret_cast<const float4*>(ctx + row * ldo + h * d)[sub];
            const float4 y = reinterpret_cast<const float4*>(dctx + row * ldo + h * d)[sub];
            acc = (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        }
#pragma unroll
        for (int o = LG / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0) {
            const int64_t nn = row / Tq, q = row % Tq;
            delta[(nn * heads + h) * Tq + q] = acc;
        }
    }
}

template <int DP, bool DROP, bool BF>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Mrow = smem + 4 * TS;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int qi = (bx * nw + wave) * 32 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const LaneOff lo = make_lane_off<DP>(l31, half);

    auto issue = [&](int t) {
        float* st = smem + (t & 1) * 2 * TS;
        Tile<DP>::issue(st, a.k, a.ldk, (int64_t)n * a.Tk, t * 32, a.Tk, col0, a.d, wave, nw, lane);
        Tile<DP>::issue(st + TS, a.v, a.ldv, (int64_t)n * a.Tk, t * 32, a.Tk, col0, a.d, wave, nw, lane);
    };
    issue(0);                                   // first K/V tile travels while the register fragments are fetched

    float Qr[DP / 2], Gr[DP / 2];
    load_rowfrag<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, a.d, half);
    load_rowfrag<DP>(Gr, a.dctx, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, a.d, half);
    const int64_t sidx = ((int64_t)n * a.heads + h) * a.Tq + qi;
    const float lse = qvalid ? a.lse[sidx] : 0.f;
    const float dl = qvalid ? a.delta[sidx] : 0.f;
    for (int j = tid; j < ntiles * 32; j += nthr) Mrow[j] = j < a.Tk ? (a.mask ? a.mask[(int64_t)n * a.Tk + j] : 0.f) : -INFINITY;

    f32x16 dQ[DP / 32];
#pragma unroll
    for (int c = 0; c < DP / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[c][r] = 0.f;

    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    const uint32_t dlo = (uint32_t)sidx;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    for (int t = 0; t < ntiles; ++t) {
        TILE_WAIT_AND_SYNC();
        if (t + 1 < ntiles) issue(t + 1);
        const float* Ks = smem + (t & 1) * 2 * TS;
        const float* Vs = Ks + TS;
        const int j0 = t * 32;

        const f32x16 S = mma_rows<DP, BF>(Ks, Qr, lo);
        const f32x16 dP = mma_rows<DP, BF>(Vs, Gr, lo);
        float dS[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mk = lds4(Mrow + j0 + 8 * g + 4 * half);
            const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float p = __expf(score(S[r], a.scale, mkv[u]) - lse);
                float dp = dP[r];
                if (DROP) dp = attn_drop_hash((uint32_t)(j0 + krow(r, half)), dlo, key) >= thr ? dp * ik : 0.f;
                dS[r] = p * (dp - dl);
            }
        }
        mma_cols<DP, BF>(dQ, Ks, dS, lo, a.d);
    }
    if (qvalid) store_cols<DP>(dQ, a.dq, a.lddq, (int64_t)n * a.Tq + qi, col0, a.d, half, a.scale);
}

template <int DP, bool DROP, bool BF>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [stage0: Q dO][stage1: Q dO][lse row, +inf past Tq][delta row]
    const int nqt = (a.Tq + 31) >> 5;
    float* Lrow = smem + 4 * TS;
    float* Drow = Lrow + nqt * 32;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int kj = (bx * nw + wave) * 32 + l31;    // this lane's key
    const bool kvalid = kj < a.Tk;
    const int col0 = h * a.d;
    const LaneOff lo = make_lane_off<DP>(l31, half);

    auto issue = [&](int t) {
        float* st = smem + (t & 1) * 2 * TS;
        Tile<DP>::issue(st, a.q, a.ldq, (int64_t)n * a.Tq, t * 32, a.Tq, col0, a.d, wave, nw, lane);
        Tile<DP>::issue(st + TS, a.dctx, a.ldo, (int64_t)n * a.Tq, t * 32, a.Tq, col0, a.d, wave, nw, lane);
    };
    issue(0);                                   // first Q/dO tile travels while the register fragments and the lse/delta rows are fetched

    float Kr[DP / 2], Vr[DP / 2];
    load_rowfrag<DP>(Kr, a.k, a.ldk, (int64_t)n * a.Tk, kj, a.Tk, col0, a.d, half);
    load_rowfrag<DP>(Vr, a.v, a.ldv, (int64_t)n * a.Tk, kj, a.Tk, col0, a.d, half);
    const float mk = kvalid ? (a.mask ? a.mask[(int64_t)n * a.Tk + kj] : 0.f) : -INFINITY;
    const int64_t srow = ((int64_t)n * a.heads + h) * a.Tq;
    for (int j = tid; j < nqt * 32; j += nthr) {
        Lrow[j] = j < a.Tq ? a.lse[srow + j] : INFINITY;
        Drow[j] = j < a.Tq ? a.delta[srow + j] : 0.f;
    }

    f32x16 dK[DP / 32], dV[DP / 32];
#pragma unroll
    for (int c = 0; c < DP / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dK[c][r] = 0.f; dV[c][r] = 0.f; }

    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    for (int t = 0; t < nqt; ++t) {
        TILE_WAIT_AND_SYNC();
        if (t + 1 < nqt) issue(t + 1);
        const float* Qs = smem + (t & 1) * 2 * TS;
        const float* Gs = Qs + TS;
        const int i0 = t * 32;

        // S[query][key] and dP[query][key]: rows = queries of the tile (krow order down the registers), column = this lane's key
        const f32x16 S = mma_rows<DP, BF>(Qs, Kr, lo);
        const f32x16 dP = mma_rows<DP, BF>(Gs, Vr, lo);
        float Pt[16], dS[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 ls = lds4(Lrow + i0 + 8 * g + 4 * half);
            const float4 ds = lds4(Drow + i0 + 8 * g + 4 * half);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dsv[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float p = __expf(score(S[r], a.scale, mk) - lsv[u]);
                float keep = 1.f;
                if (DROP) keep = attn_drop_hash((uint32_t)kj, (uint32_t)(srow + i0 + 8 * g + 4 * half + u), key) >= thr ? ik : 0.f;
                Pt[r] = p * keep;
                dS[r] = p * (dP[r] * keep - dsv[u]);
            }
        }
        mma_cols<DP, BF>(dV, Gs, Pt, lo, a.d);
        mma_cols<DP, BF>(dK, Qs, dS, lo, a.d);
    }
    if (kvalid) {
        store_cols<DP>(dV, a.dv, a.lddv, (int64_t)n * a.Tk + kj, col0, a.d, half, 1.0f);
        store_cols<DP>(dK, a.dk, a.lddk, (int64_t)n * a.Tk + kj, col0, a.d, half, a.scale);
    }
}

// diagnostic: materialise attention_probs (reference returns them when output_all_attention_masks=True)
__global__ __launch_bounds__(256) void attn_probs_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                         int64_t ldk, const float* __restrict__ mask, const float* __restrict__ lse,
                                                         float* __restrict__ probs, int N, int heads, int Tq, int Tk, int d,
                                                         float scale) {
    const int64_t total = (int64_t)N * heads * Tq * Tk;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % Tk);
        int64_t t = i / Tk;
        const int qi = (int)(t % Tq); t /= Tq;
        const int h = (int)(t % heads);
        const int64_t n = t / heads;
        const float* qp = q + (n * Tq + qi) * ldq + h * d;
        const float* kp = k + (n * Tk + j) * ldk + h * d;
        float acc = 0.f;
        for (int c = 0; c < d; ++c) acc = fmaf(qp[c], kp[c], acc);
        probs[i] = __expf(score(acc, scale, mask ? mask[n * Tk + j] : 0.f) - lse[(n * heads + h) * Tq + qi]);
    }
}

// ---- kernel entry points: one problem per launch (3-D grid), or the two directions of BertBiAttention in ONE launch (1-D grid:
// the first nb0 workgroups belong to direction 0).  The pair launch lets the hardware dispatcher fill the slots that one
// direction's last partial round would leave idle with the other direction's workgroups.
struct AttnPair { AttnArgs p[2]; int nb0, gx0, gx1; };

#define YT_ATTN_KERNELS(NAME, BODY, LB)                                                                                  \
    template <int DP, bool DROP, bool BF>                                                                                \
    __global__ LB void NAME##_kernel(const AttnArgs a) { BODY<DP, DROP, BF>(a, blockIdx.x, blockIdx.y, blockIdx.z); }    \
    template <int DP, bool DROP, bool BF>                                                                                \
    __global__ LB void NAME##_pair_kernel(const AttnPair b) {                                                            \
        int bid = blockIdx.x;                                                                                            \
        if (bid < b.nb0) {                                                                                               \
            BODY<DP, DROP, BF>(b.p[0], bid % b.gx0, (bid / b.gx0) % b.p[0].heads, bid / (b.gx0 * b.p[0].heads));         \
        } else {                                                                                                         \
            bid -= b.nb0;                                                                                                \
            BODY<DP, DROP, BF>(b.p[1], bid % b.gx1, (bid / b.gx1) % b.p[1].heads, bid / (b.gx1 * b.p[1].heads));         \
        }                                                                                                                \
    }
YT_ATTN_KERNELS(attn_fwd, attn_fwd_body, __launch_bounds__(256, 2))
YT_ATTN_KERNELS(attn_bwd_dq, attn_bwd_dq_body, __launch_bounds__(256, 2))
YT_ATTN_KERNELS(attn_bwd_dkv, attn_bwd_dkv_body, __launch_bounds__(256))
#undef YT_ATTN_KERNELS


static int pick_waves(int T) {
    // waves per workgroup (32 rows each): least padding first, then the most waves (they share the staged tiles)
    static const int force = getenv("YTVLN_ATTN_WAVES") ? atoi(getenv("YTVLN_ATTN_WAVES")) : 0;      // experiment knob
    if (force >= 1 && force <= 4) return force;
    int best = 1, best_pad = 1 << 30;
    for (int nw = 4; nw >= 1; --nw) {
        const int pad = (int)cdiv(T, 32 * nw) * 32 * nw - T;
        if (pad < best_pad) { best_pad = pad; best = nw; }
    }
    return best;
}

static int check_common(const char* who, const AttnArgs& a) {
    YT_REQUIRE(a.q && a.k && a.v, "%s: null q/k/v", who);
    YT_REQUIRE(a.N > 0 && a.heads > 0 && a.Tq > 0 && a.Tk > 0, "%s: empty problem", who);
    YT_REQUIRE(a.d > 0 && a.d % 4 == 0 && a.d <= 128, "%s: head dim %d unsupported (multiple of 4, <= 128)", who, a.d);
    YT_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0, "%s: leading dimensions must be multiples of 4", who);
    YT_REQUIRE(((uintptr_t)a.q & 15) == 0 && ((uintptr_t)a.k & 15) == 0 && ((uintptr_t)a.v & 15) == 0, "%s: q/k/v must be 16-byte aligned", who);
    YT_REQUIRE(a.p_drop >= 0.f && a.p_drop < 1.f, "%s: p_drop out of range", who);
    YT_REQUIRE(!(a.p_drop > 0.f) || a.rng, "%s: dropout needs rng state", who);
    YT_REQUIRE(a.N <= 65535 && a.heads <= 65535, "%s: grid too large", who);
    return 0;
}

#define DISPATCH_DP_DROP_BF(KERNEL, BFV, grid, block, lds_fn, s, a)                                            \
    do {                                                                                                     \
        const bool drop_ = (a).p_drop > 0.f;                                                                 \
        if ((a).d <= 32) {                                                                                   \
            if (drop_) hipLaunchKernelGGL((KERNEL<32, true, BFV>), grid, block, lds_fn(32), s, a);           \
            else hipLaunchKernelGGL((KERNEL<32, false, BFV>), grid, block, lds_fn(32), s, a);                \
        } else if ((a).d <= 64) {                                                                            \
            if (drop_) hipLaunchKernelGGL((KERNEL<64, true, BFV>), grid, block, lds_fn(64), s, a);           \
            else hipLaunchKernelGGL((KERNEL<64, false, BFV>), grid, block, lds_fn(64), s, a);                \
        } else {                                                                                             \
            if (drop_) hipLaunchKernelGGL((KERNEL<128, true, BFV>), grid, block, lds_fn(128), s, a);         \
            else hipLaunchKernelGGL((KERNEL<128, false, BFV>), grid, block, lds_fn(128), s, a);              \
        }                                                                                                    \
    } while (0)
#define DISPATCH_DP_DROP(KERNEL, grid, block, lds_fn, s, a)                           \
    do {                                                                              \
        if ((a).bf16) DISPATCH_DP_DROP_BF(KERNEL, true, grid, block, lds_fn, s, a);   \
        else DISPATCH_DP_DROP_BF(KERNEL, false, grid, block, lds_fn, s, a);           \
    } while (0)

// dynamic LDS: the two-stage tile ring + the streamed dimension's mask row (forward / dQ) or lse and delta rows (dK/dV), staged whole
struct LdsFwd { int rows; size_t operator()(int dp) const { return (size_t)(4 * 32 * dp + ((rows + 31) / 32) * 32) * sizeof(float); } };
struct LdsBwd { int rows; size_t operator()(int dp) const { return (size_t)(4 * 32 * dp + 2 * ((rows + 31) / 32) * 32) * sizeof(float); } };

}  // namespace ytvln

using namespace ytvln;

static void launch_delta(const float* ctx, const float* dctx, int64_t ldo, float* delta, int N, int heads, int Tq, int d, hipStream_t s) {
    const int64_t total = (int64_t)N * Tq * heads;
    {
        const int d4 = d / 4;
        const int lg = d4 <= 1 ? 1 : d4 <= 2 ? 2 : d4 <= 4 ? 4 : d4 <= 8 ? 8 : d4 <= 16 ? 16 : 32;
        const dim3 dgrid((unsigned)std::min<int64_t>(cdiv(total * lg, 256), 8192));
#define YT_DELTA(L) hipLaunchKernelGGL(attn_delta_kernel<L>, dgrid, dim3(256), 0, s, ctx, dctx, ldo, delta, N, heads, Tq, d)
        switch (lg) {
            case 1: YT_DELTA(1); break;
            case 2: YT_DELTA(2); break;
            case 4: YT_DELTA(4); break;
            case 8: YT_DELTA(8); break;
            case 16: YT_DELTA(16); break;
            default: YT_DELTA(32); break;
        }
#undef YT_DELTA
    }
}

static int attn_fwd_impl(int bf16, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                         const float* mask, float* ctx, int64_t ldo, float* lse, int N, int heads, int Tq, int Tk,
                         int d, float scale, float p_drop, const int64_t* rng, int64_t site, void* stream) {
    AttnArgs a = {};
    a.bf16 = bf16;
    a.q = q; a.k = k; a.v = v; a.mask = mask; a.out = ctx; a.lse_out = lse;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.N = N; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d; a.scale = scale; a.p_drop = p_drop; a.rng = rng; a.site = site;
    if (int rc = check_common("attn_fwd", a)) return rc;
    YT_REQUIRE(ctx && lse && ((uintptr_t)ctx & 15) == 0, "attn_fwd: ctx/lse null or misaligned");
    const int nw = pick_waves(Tq);
    dim3 grid((unsigned)cdiv(Tq, 32 * nw), heads, N), block(64 * nw);
    hipStream_t s = as_stream(stream);
    YT_REQUIRE(Tk <= 8192 && Tq <= 8192, "attn_fwd: sequence too long for the LDS-resident mask row");
    const LdsFwd lds_fwd{Tk};
    DISPATCH_DP_DROP(attn_fwd_kernel, grid, block, lds_fwd, s, a);
    YT_LAUNCH_CHECK("attn_fwd");
    return 0;
}

extern "C" int ytvln_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                  const float* mask, float* ctx, int64_t ldo, float* lse, int N, int heads, int Tq, int Tk,
                                  int d, float scale, float p_drop, const int64_t* rng, int64_t site, void* stream) {
    return attn_fwd_impl(0, q, ldq, k, ldk, v, ldv, mask, ctx, ldo, lse, N, heads, Tq, Tk, d, scale, p_drop, rng, site, stream);
}

extern "C" int ytvln_attn_fwd_bf16(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                   const float* mask, float* ctx, int64_t ldo, float* lse, int N, int heads, int Tq, int Tk,
                                   int d, float scale, float p_drop, const int64_t* rng, int64_t site, void* stream) {
    YT_REQUIRE(d % 8 == 0, "attn_fwd_bf16: head dim %d must be a multiple of 8", d);
    return attn_fwd_impl(1, q, ldq, k, ldk, v, ldv, mask, ctx, ldo, lse, N, heads, Tq, Tk, d, scale, p_drop, rng, site, stream);
}

static int attn_bwd_impl(int bf16, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                         const float* mask, const float* ctx, const float* dctx, int64_t ldo, const float* lse,
                         float* delta, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                         int N, int heads, int Tq, int Tk, int d, float scale, float p_drop, const int64_t* rng,
                         int64_t site, void* stream) {
    AttnArgs a = {};
    a.bf16 = bf16;
    a.q = q; a.k = k; a.v = v; a.mask = mask; a.ctx = ctx; a.dctx = dctx; a.lse = lse; a.delta = delta;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.N = N; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d; a.scale = scale; a.p_drop = p_drop; a.rng = rng; a.site = site;
    if (int rc = check_common("attn_bwd", a)) return rc;
    YT_REQUIRE(ctx && dctx && lse && delta && dq && dk && dv, "attn_bwd: null pointer");
    YT_REQUIRE(Tk <= 8192 && Tq <= 8192, "attn_bwd: sequence too long for the LDS-resident mask / lse rows");
    YT_REQUIRE(lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0, "attn_bwd: gradient leading dimensions must be multiples of 4");
    YT_REQUIRE((((uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0, "attn_bwd: misaligned pointer");
    hipStream_t s = as_stream(stream);
    const int64_t total = (int64_t)N * Tq * heads;
    launch_delta(ctx, dctx, ldo, delta, N, heads, Tq, d, s);
    {
        const int nw = pick_waves(Tq);
        dim3 grid((unsigned)cdiv(Tq, 32 * nw), heads, N), block(64 * nw);
        const LdsFwd lds_fwd{Tk};
        DISPATCH_DP_DROP(attn_bwd_dq_kernel, grid, block, lds_fwd, s, a);
    }
    {
        const int nw = pick_waves(Tk);
        dim3 grid((unsigned)cdiv(Tk, 32 * nw), heads, N), block(64 * nw);
        const LdsBwd lds_bwd{Tq};
        DISPATCH_DP_DROP(attn_bwd_dkv_kernel, grid, block, lds_bwd, s, a);
    }
    YT_LAUNCH_CHECK("attn_bwd");
    return 0;
}

extern "C" int ytvln_attn_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                  const float* mask, const float* ctx, const float* dctx, int64_t ldo, const float* lse,
                                  float* delta, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                                  int N, int heads, int Tq, int Tk, int d, float scale, float p_drop, const int64_t* rng,
                                  int64_t site, void* stream) {
    return attn_bwd_impl(0, q, ldq, k, ldk, v, ldv, mask, ctx, dctx, ldo, lse, delta, dq, lddq, dk, lddk, dv, lddv, N, heads, Tq, Tk, d,
                         scale, p_drop, rng, site, stream);
}

extern "C" int ytvln_attn_bwd_bf16(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                   const float* mask, const float* ctx, const float* dctx, int64_t ldo, const float* lse,
                                   float* delta, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                                   int N, int heads, int Tq, int Tk, int d, float scale, float p_drop, const int64_t* rng,
                                   int64_t site, void* stream) {
    YT_REQUIRE(d % 8 == 0, "attn_bwd_bf16: head dim %d must be a multiple of 8", d);
    return attn_bwd_impl(1, q, ldq, k, ldk, v, ldv, mask, ctx, dctx, ldo, lse, delta, dq, lddq, dk, lddk, dv, lddv, N, heads, Tq, Tk, d,
                         scale, p_drop, rng, site, stream);
}

// ---- both directions of BertBiAttention in one launch -----------------------------------------------------------------------
static void fill_args(AttnArgs& a, const ytvln_attn_problem& pr, int N, int heads, int d, float scale, const int64_t* rng, int bf16) {
    a = AttnArgs{};
    a.q = pr.q; a.k = pr.k; a.v = pr.v; a.mask = pr.mask;
    a.ctx = pr.ctx_in; a.dctx = pr.dctx; a.lse = pr.lse_in; a.delta = pr.delta;
    a.out = pr.ctx; a.lse_out = pr.lse; a.dq = pr.dq; a.dk = pr.dk; a.dv = pr.dv;
    a.ldq = pr.ldq; a.ldk = pr.ldk; a.ldv = pr.ldv; a.ldo = pr.ldo; a.lddq = pr.lddq; a.lddk = pr.lddk; a.lddv = pr.lddv;
    a.N = N; a.heads = heads; a.Tq = pr.Tq; a.Tk = pr.Tk; a.d = d; a.scale = scale; a.p_drop = pr.p_drop; a.rng = rng; a.site = pr.site;
    a.bf16 = bf16;
}

#define DISPATCH_PAIR_BF(KERNEL, BFV, drop_, grid, block, lds, s, b)                                           \
    do {                                                                                                     \
        if ((b).p[0].d <= 32) {                                                                              \
            if (drop_) hipLaunchKernelGGL((KERNEL<32, true, BFV>), grid, block, lds(32), s, b);              \
            else hipLaunchKernelGGL((KERNEL<32, false, BFV>), grid, block, lds(32), s, b);                   \
        } else if ((b).p[0].d <= 64) {                                                                       \
            if (drop_) hipLaunchKernelGGL((KERNEL<64, true, BFV>), grid, block, lds(64), s, b);              \
            else hipLaunchKernelGGL((KERNEL<64, false, BFV>), grid, block, lds(64), s, b);                   \
        } else {                                                                                             \
            if (drop_) hipLaunchKernelGGL((KERNEL<128, true, BFV>), grid, block, lds(128), s, b);            \
            else hipLaunchKernelGGL((KERNEL<128, false, BFV>), grid, block, lds(128), s, b);                 \
        }                                                                                                    \
    } while (0)
#define DISPATCH_PAIR(KERNEL, drop_, grid, block, lds, s, b)                              \
    do {                                                                                  \
        if ((b).p[0].bf16) DISPATCH_PAIR_BF(KERNEL, true, drop_, grid, block, lds, s, b); \
        else DISPATCH_PAIR_BF(KERNEL, false, drop_, grid, block, lds, s, b);              \
    } while (0)

extern "C" int ytvln_attn_fwd_pair(const ytvln_attn_problem* pa, const ytvln_attn_problem* pb, int N, int heads, int d, float scale,
                                   const int64_t* rng, int bf16, void* stream) {
    YT_REQUIRE(pa && pb, "attn_fwd_pair: null problem");
    const int nwa = pick_waves(pa->Tq), nwb = pick_waves(pb->Tq);
    const bool dropa = pa->p_drop > 0.f, dropb = pb->p_drop > 0.f;
    if (nwa != nwb || (bf16 && d % 8 != 0)) {       // different workgroup shapes: two ordinary launches
        const ytvln_attn_problem* ps[2] = {pa, pb};
        for (const ytvln_attn_problem* p : ps)
            if (int rc = attn_fwd_impl(bf16 && d % 8 == 0, p->q, p->ldq, p->k, p->ldk, p->v, p->ldv, p->mask, p->ctx, p->ldo, p->lse, N, heads,
                                       p->Tq, p->Tk, d, scale, p->p_drop, rng, p->site, stream))
                return rc;
        return 0;
    }
    AttnPair b;
    fill_args(b.p[0], *pa, N, heads, d, scale, rng, bf16);
    fill_args(b.p[1], *pb, N, heads, d, scale, rng, bf16);
    for (int i = 0; i < 2; ++i) {
        if (int rc = check_common("attn_fwd_pair", b.p[i])) return rc;
        YT_REQUIRE(b.p[i].out && b.p[i].lse_out && ((uintptr_t)b.p[i].out & 15) == 0, "attn_fwd_pair: ctx/lse null or misaligned");
        YT_REQUIRE(b.p[i].Tk <= 8192 && b.p[i].Tq <= 8192, "attn_fwd_pair: sequence too long for the LDS-resident mask row");
    }
    b.gx0 = (int)cdiv(pa->Tq, 32 * nwa); b.gx1 = (int)cdiv(pb->Tq, 32 * nwa);
    b.nb0 = b.gx0 * heads * N;
    const int64_t total = (int64_t)b.nb0 + (int64_t)b.gx1 * heads * N;
    YT_REQUIRE(total < (1ll << 31), "attn_fwd_pair: grid too large");
    const LdsFwd lds{std::max(pa->Tk, pb->Tk)};
    hipStream_t s = as_stream(stream);
    DISPATCH_PAIR(attn_fwd_pair_kernel, (dropa || dropb), dim3((unsigned)total), dim3(64 * nwa), lds, s, b);
    YT_LAUNCH_CHECK("attn_fwd_pair");
    return 0;
}

extern "C" int ytvln_attn_bwd_pair(const ytvln_attn_problem* pa, const ytvln_attn_problem* pb, int N, int heads, int d, float scale,
                                   const int64_t* rng, int bf16, void* stream) {
    YT_REQUIRE(pa && pb, "attn_bwd_pair: null problem");
    const bool same = pick_waves(pa->Tq) == pick_waves(pb->Tq) && pick_waves(pa->Tk) == pick_waves(pb->Tk) && !(bf16 && d % 8 != 0);
    if (!same) {
        const ytvln_attn_problem* ps[2] = {pa, pb};
        for (const ytvln_attn_problem* p : ps)
            if (int rc = attn_bwd_impl(bf16 && d % 8 == 0, p->q, p->ldq, p->k, p->ldk, p->v, p->ldv, p->mask, p->ctx_in, p->dctx, p->ldo, p->lse_in,
                                       p->delta, p->dq, p->lddq, p->dk, p->lddk, p->dv, p->lddv, N, heads, p->Tq, p->Tk, d, scale, p->p_drop,
                                       rng, p->site, stream))
                return rc;
        return 0;
    }
    AttnPair b;
    fill_args(b.p[0], *pa, N, heads, d, scale, rng, bf16);
    fill_args(b.p[1], *pb, N, heads, d, scale, rng, bf16);
    hipStream_t s = as_stream(stream);
    for (int i = 0; i < 2; ++i) {
        const AttnArgs& a = b.p[i];
        if (int rc = check_common("attn_bwd_pair", a)) return rc;
        YT_REQUIRE(a.ctx && a.dctx && a.lse && a.delta && a.dq && a.dk && a.dv, "attn_bwd_pair: null pointer");
        YT_REQUIRE(a.Tk <= 8192 && a.Tq <= 8192, "attn_bwd_pair: sequence too long for the LDS-resident mask / lse rows");
        YT_REQUIRE(a.lddq % 4 == 0 && a.lddk % 4 == 0 && a.lddv % 4 == 0, "attn_bwd_pair: gradient leading dimensions must be multiples of 4");
        YT_REQUIRE((((uintptr_t)a.ctx | (uintptr_t)a.dctx | (uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv) & 15) == 0, "attn_bwd_pair: misaligned pointer");
        launch_delta(a.ctx, a.dctx, a.ldo, (i == 0 ? pa : pb)->delta, N, heads, a.Tq, d, s);
    }
    const bool drop = pa->p_drop > 0.f || pb->p_drop > 0.f;
    {
        const int nw = pick_waves(pa->Tq);
        b.gx0 = (int)cdiv(pa->Tq, 32 * nw); b.gx1 = (int)cdiv(pb->Tq, 32 * nw);
        b.nb0 = b.gx0 * heads * N;
        const int64_t total = (int64_t)b.nb0 + (int64_t)b.gx1 * heads * N;
        YT_REQUIRE(total < (1ll << 31), "attn_bwd_pair: grid too large");
        const LdsFwd lds{std::max(pa->Tk, pb->Tk)};
        DISPATCH_PAIR(attn_bwd_dq_pair_kernel, drop, dim3((unsigned)total), dim3(64 * nw), lds, s, b);
    }
    {
        const int nw = pick_waves(pa->Tk);
        b.gx0 = (int)cdiv(pa->Tk, 32 * nw); b.gx1 = (int)cdiv(pb->Tk, 32 * nw);
        b.nb0 = b.gx0 * heads * N;
        const int64_t total = (int64_t)b.nb0 + (int64_t)b.gx1 * heads * N;
        YT_REQUIRE(total < (1ll << 31), "attn_bwd_pair: grid too large");
        const LdsBwd lds{std::max(pa->Tq, pb->Tq)};
        DISPATCH_PAIR(attn_bwd_dkv_pair_kernel, drop, dim3((unsigned)total), dim3(64 * nw), lds, s, b);
    }
    YT_LAUNCH_CHECK("attn_bwd_pair");
    return 0;
}

extern "C" int ytvln_attn_probs_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* mask, const float* lse,
                                    float* probs, int N, int heads, int Tq, int Tk, int d, float scale, void* stream) {
    YT_REQUIRE(q && k && lse && probs, "attn_probs: null pointer");
    const int64_t total = (int64_t)N * heads * Tq * Tk;
    if (total == 0) return 0;
    hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 8192)), dim3(256), 0, as_stream(stream), q, ldq,
                       k, ldk, mask, lse, probs, N, heads, Tq, Tk, d, scale);
    YT_LAUNCH_CHECK("attn_probs");
    return 0;
}
