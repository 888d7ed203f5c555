// Fused multi-head attention for the bf16-resident path (BASELINE configs[4]): bf16 q / k / v / ctx and gradients in HBM, every contraction on
// v_mfma_f32_32x32x16_bf16, fp32 softmax / log-sum-exp / delta.  Same arithmetic as attention.hip (BertSelfAttention, BertImageSelfAttention
// and both directions of BertBiAttention, vilbert/vilbert.py:284-311, 413-440, 552-618; backward = recompute-based flash backward) with the
// matrix work 16x shorter, so the kernels are shaped around what is left -- the softmax / dropout VALU work and the operand path:
//
//   * ONE wave per workgroup owning 32 queries (forward, dQ) or 32 keys (dK/dV), <= 256 registers and ~17 KB of LDS, so TWO waves share a SIMD
//     and one wave's matrix instructions run under the other's softmax arithmetic; no barriers, the only waits are the wave's own counted vmcnt;
//   * everything is computed TRANSPOSED so that the reduction axes are lane-local and nothing is ever permuted across lanes:
//        S^T[key][query] = K . Q^T          (query on the lane: row max / row sum = per-lane loops + one half-wave swap)
//        O^T[d][query]  += V^T . P^T        (the P registers of a lane ARE its B operand; the per-query rescale and 1/l are lane-local scalars)
//     and likewise dQ^T += K^T . dS^T, dV^T += dO^T . P, dK^T += Q^T . dS in the backward kernels;
//   * K / V (Q / dO in the dK/dV kernel) tiles of 32 rows x d bf16 arrive by LDS-DMA in their HBM layout, 16-byte granules XOR-swizzled (on the
//     per-lane SOURCE address) so that BOTH ways a tile is read are bank-conflict free: ds_read_b128 of a lane's own row (A operand of K.Q^T,
//     contraction over d) and ds_read_b64_tr_b16 of [4 rows][16 columns] blocks (A operand of V^T.P^T, contraction over the rows: the hardware
//     transpose delivers the 8 rows a lane feeds to the matrix instruction -- no transposed copy of V, K, Q or dO exists anywhere);
//   * outputs leave as bf16, 8 bytes per lane (4 consecutive head columns of the lane's own row).
// Dropout of the probabilities: the same per-score hash as attention.hip (common.h: attn_drop_hash), drawn ONCE, in the forward kernel.  The
// keep decisions are the forward's compare results -- for accumulator register r of a (32 queries x 32 keys) block a 64-bit lane mask (lane =
// query + 32 half, key = bkrow(r, half)) -- and they are written out as such: 16 masks = 128 bytes per block.  dQ has the forward's lane layout
// and reads a block's masks with scalar loads straight into the select operand of v_cndmask; dK/dV (key on the lane, queries down the
// registers) reads ONE 8-byte mask per lane and block and tests a bit per score.  (Round 4 regenerated the hash in both backward kernels:
// ~10 of the ~20 vector instructions per score there.)
// Head dimensions 64 and 128 (unpadded); sequences up to 8192 keys / queries (the mask / lse rows live in LDS).
#include "common.h"
#include <algorithm>

namespace ytvln {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t bf16_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct BAttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const float* mask;
    const bf16_t* ctx; const bf16_t* dctx; const float* lse; const float* delta;
    bf16_t* out; float* lse_out; bf16_t* dq; bf16_t* dk; bf16_t* dv; float* delta_out;
    int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int N, heads, Tq, Tk, d;
    float scale, p_drop;
    const int64_t* rng; int64_t site;
    uint64_t* keep;              // dropout keep decisions of the forward: [N * heads][query block][key tile][16] 64-bit lane masks (see battn_fwd_body)
};
struct BAttnLaunch { BAttnArgs p[2]; int nb0, gx0, gx1; };

#define MFMA_B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define B_RESCALE_THR 12.0f

__device__ __forceinline__ int bkrow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ float bscore(float s, float scale, float mask) { return __fadd_rn(__fmul_rn(s, scale), mask); }
// Before a tile buffer is handed back to the LDS-DMA: every LDS read of it must have RETURNED, not merely been issued.  The reads feed matrix
// instructions, and hipcc is free to sink those (and the s_waitcnt lgkmcnt in front of them) below the next global_load_lds: the DMA of a
// tile that hits in L1 / L2 (the other query blocks of the same head load the same keys) then overtakes reads still queued behind the other
// waves of the CU and a whole 32-row block of the output is computed from a half-replaced tile (seen as 1-3 wrong blocks in ~3500 per launch,
// tools/bf16_repro.py).  A compiler barrier is not enough; this is.
__device__ __forceinline__ void b_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int N>
__device__ __forceinline__ void b_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ bf16x8 bpack8(const float* p) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)p[e];
    return v;
}

// ---- a 32-row x DP bf16 tile in LDS ------------------------------------------------------------------------------------------------
// Row r at byte r * 2 DP; 16-byte granule g of row r stored at position g ^ swz(r).  The swizzle is chosen so that (checked by brute force over
// the lane groups of ds_read_b128 and ds_read_b64): a wave reading granule g of rows lane & 31, and a wave reading [4 rows][16 columns] blocks for
// the transposing read, both touch every LDS bank once per group.
template <int DP>
struct BTile {
    static constexpr int ROWB = DP * 2, GPR = DP / 8, PC = DP / 16, RPP = 1024 / ROWB;      // bytes per row, granules per row, 1 KiB pieces per tile, rows per piece
    static constexpr int BYTES = 32 * ROWB;
    __device__ static __forceinline__ int swz(int row) {
        return DP == 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
    }
    // A operand with the contraction over the head dimension: this lane's row (lane & 31), d = 16 s + 8 half + 0..7
    __device__ static __forceinline__ bf16x8 kc(const char* __restrict__ T, int l31, int half, int s) {
        return *reinterpret_cast<const bf16x8*>(T + l31 * ROWB + (((2 * s + half) ^ swz(l31)) << 4));
    }
    // A operand with the contraction over the tile's rows (the tile read TRANSPOSED): column 32 c + (lane & 31), rows 16 m + 8 (e >> 2) + 4 half
    // + (e & 3) for e = 0..7 -- the order in which a lane's score registers hold those rows (bkrow), so P / dS feed the B operand as they are
    __device__ static __forceinline__ bf16x8 tr(const char* __restrict__ T, int lane, int m, int c) {
        const int h = lane >> 5, r = (lane & 15) >> 2, c4 = lane & 3, blk = (lane >> 4) & 1;
        const int col = 32 * c + 16 * blk + 4 * c4;
        s16x4 v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 16 * m + 8 * t + 4 * h + r;
            const char* p = T + row * ROWB + (((col >> 3) ^ swz(row)) << 4) + ((col & 7) << 1);
            v[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_ptr_t)(const_cast<char*>(p)));
        }
        const s16x8 w = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
        return __builtin_bit_cast(bf16x8, w);
    }
};

// One LDS-DMA stream: 32-row tiles of a row-major bf16 matrix (this (pair, head)'s column block) into one LDS buffer, PC pieces of 1 KiB.
// Rows past the end are clamped to the last row (finite data): such keys carry a -inf mask, such queries a +inf lse, and their outputs are
// not stored.
template <int DP>
struct BStream {
    using T = BTile<DP>;
    const bf16_t* __restrict__ base;
    lds_char* lds;
    int ld, nrows;
    int prow[T::PC], pcol[T::PC];          // this lane's row inside the tile / source column (elements) for each piece
    __device__ __forceinline__ void init(const bf16_t* b, char* ldsbuf, int ld_, int nrows_, int lane) {
        base = b; lds = (lds_char*)(lds_ptr_t)ldsbuf; ld = ld_; nrows = nrows_;
#pragma unroll
        for (int p = 0; p < T::PC; ++p) {
            const int idx = p * 64 + lane, row = idx / T::GPR, pg = idx % T::GPR;
            prow[p] = row;
            pcol[p] = 8 * (pg ^ T::swz(row));
        }
    }
    __device__ __forceinline__ void issue(const int row0) const {
        static_for<T::PC>([&](auto PT) __attribute__((always_inline)) {
            constexpr int p = decltype(PT)::value;
            const int row = min(row0 + prow[p], nrows - 1);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (uint32_t)(row * ld + pcol[p])), (lds_ptr_t)(lds + p * 1024), 16, 0, 0);
        });
    }
};

// this lane's row fragment from global memory: X[row][16 s + 8 half + 0..7], s = 0 .. DP/16-1 (a row past the end repeats the last one)
template <int DP>
__device__ __forceinline__ void bload_rowfrag(bf16x8 (&R)[DP / 16], const bf16_t* __restrict__ base, int64_t ld, int64_t row_base, int row, int nrows,
                                              int col0, int half) {
    const bf16_t* __restrict__ rp = base + (row_base + min(row, nrows - 1)) * ld + col0 + 8 * half;
#pragma unroll
    for (int s = 0; s < DP / 16; ++s) R[s] = *reinterpret_cast<const bf16x8*>(rp + 16 * s);
}

// acc^T[c][r] = value at (row = this lane's row, column 32 c + bkrow(r, half)) -> bf16, 8 bytes (4 consecutive columns) per store
template <int DP>
__device__ __forceinline__ void bstore_rows(const f32x16 (&acc)[DP / 32], bf16_t* __restrict__ base, int64_t ld, int64_t row_base, int row, int nrows,
                                            int col0, int half, float mul) {
    if (row >= nrows) return;
    bf16_t* rp = base + (row_base + row) * ld + col0 + 4 * half;
#pragma unroll
    for (int c = 0; c < DP / 32; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t lo = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(acc[c][4 * q] * mul)) |
                                ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(acc[c][4 * q + 1] * mul)) << 16);
            const uint32_t hi = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(acc[c][4 * q + 2] * mul)) |
                                ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(acc[c][4 * q + 3] * mul)) << 16);
            *reinterpret_cast<uint2*>(rp + 32 * c + 8 * q) = make_uint2(lo, hi);
        }
}

__device__ __forceinline__ float4 blds4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }

// dst[j] = j < n ? (src ? src[j] : fill_in) : fill_out for j < n32 (a multiple of 32), four loads in flight per lane (a plain loop would pay one
// memory round trip per 64 elements)
__device__ __forceinline__ void bfill_row(float* __restrict__ dst, const float* __restrict__ src, int n, int n32, float fill_in, float fill_out, int lane) {
    for (int j0 = 0; j0 < n32; j0 += 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 64 * u + lane;
            v[u] = (src && j < n) ? src[j] : fill_in;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 64 * u + lane;
            if (j < n32) dst[j] = j < n ? v[u] : fill_out;
        }
    }
}

__device__ __forceinline__ float b_halves_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float b_halves_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- forward -------------------------------------------------------------------------------------------------------------------------
// LDS: [K tile][V tile][mask row, -inf past Tk].  Per key tile t:
//     wait K(t)   S^T = K(t).Q^T          -> DMA K(t+1)
//     softmax (online, lazily moved reference), dropout, P -> bf16
//     wait V(t)   O^T += V(t)^T.P^T       -> DMA V(t+1)
template <int DP, bool DROP>
__device__ __forceinline__ void battn_fwd_body(const BAttnArgs& a, const int bx, const int h, const int n) {
    using T = BTile<DP>;
    constexpr int NS = DP / 16, NC = DP / 32, PC = T::PC;
    extern __shared__ __attribute__((aligned(16))) char bsmem[];
    char* __restrict__ Ks = bsmem;
    char* __restrict__ Vs = bsmem + T::BYTES;
    float* __restrict__ Mrow = reinterpret_cast<float*>(bsmem + 2 * T::BYTES);
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int q0 = bx * 32, qi = q0 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const int64_t krow_base = (int64_t)n * a.Tk;
    BStream<DP> ks, vs;
    ks.init(a.k + krow_base * a.ldk + col0, Ks, (int)a.ldk, a.Tk, lane);
    vs.init(a.v + krow_base * a.ldv + col0, Vs, (int)a.ldv, a.Tk, lane);

    // Prologue: the two DMA tiles first, then the register fragment, the small loads (mask row, dropout key) LAST -- vector memory retires in
    // order, so by the time the mask row is in LDS everything before it has landed: one memory round trip for the whole prologue.
    ks.issue(0);
    vs.issue(0);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 Qr[NS];
    bload_rowfrag<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
    __builtin_amdgcn_sched_barrier(0);
    bfill_row(Mrow, a.mask ? a.mask + krow_base : nullptr, a.Tk, ntiles * 32, 0.f, -INFINITY, lane);
    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    f32x16 O[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const uint32_t dlo = (uint32_t)((((int64_t)n * a.heads + h) * a.Tq + qi));
    // (a launch with dropout in either of its problems carries a keep buffer for both: the host checks it)
    uint64_t* const kblock = DROP ? a.keep + (((int64_t)n * a.heads + h) * ((a.Tq + 31) >> 5) + bx) * ntiles * 16 : nullptr;

    auto tile = [&](auto FIRST_T, auto MORE_T, const int t) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(FIRST_T)::value, more = decltype(MORE_T)::value;
        const int j0 = t * 32;
        if (FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // K(0), V(0), Q
        else b_wait<PC>();                                                  // K(t); V(t) may still be on its way
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) S = MFMA_B(T::kc(Ks, l31, half, s), Qr[s], S);
        b_reads_done();
        if (more) ks.issue(j0 + 32);
        float P[16];
        float mt = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mk = blds4(Mrow + j0 + 8 * g + 4 * half);
            P[4 * g] = bscore(S[4 * g], a.scale, mk.x); P[4 * g + 1] = bscore(S[4 * g + 1], a.scale, mk.y);
            P[4 * g + 2] = bscore(S[4 * g + 2], a.scale, mk.z); P[4 * g + 3] = bscore(S[4 * g + 3], a.scale, mk.w);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, P[r]);
        mt = b_halves_max(mt);
        if (__any(mt > m + B_RESCALE_THR)) {          // lazily moved softmax reference: O and l are rescaled only when a row maximum grew by > THR
            const float mn = fmaxf(m, mt);
            const float alpha = __expf(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[c][r] *= alpha;          // (the query is on the lane: a lane-local scalar)
        }
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { P[r] = __expf(P[r] - m); ps += P[r]; }
        l += b_halves_sum(ps);
        if (DROP) {
            uint32_t mlo = 0, mhi = 0;          // lane r (< 16) collects mask r
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const bool kp = attn_drop_hash((uint32_t)(j0 + bkrow(r, half)), dlo, key) >= thr;
                const uint64_t m = __ballot(kp);
                P[r] = kp ? P[r] * ik : 0.f;
                uint32_t lo = mlo, hi = mhi;
                // (one scalar operand per vector instruction: the lane is an inline constant.  s_nop: the mask is a scalar the VALU has just written
                //  (v_cmp); gfx950 wants two wait states before another vector instruction reads it, and the hazard recogniser does not see into
                //  inline asm -- without them the low words of most masks of a tile came out wrong, 7 % of the stored bits)
                asm("s_nop 3\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "n"(r));
                mlo = lo; mhi = hi;
            });
            if (lane < 16) kblock[(int64_t)t * 16 + lane] = ((uint64_t)mhi << 32) | mlo;
        }
        const bf16x8 Pb[2] = {bpack8(P), bpack8(P + 8)};
        if (more) b_wait<PC>();                                              // V(t); K(t+1) may still be on its way
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int c = 0; c < NC; ++c) O[c] = MFMA_B(T::tr(Vs, lane, mm, c), Pb[mm], O[c]);
        b_reads_done();
        if (more) vs.issue(j0 + 32);
    };
    if (ntiles == 1) {
        tile(std::true_type{}, std::false_type{}, 0);
    } else {
        tile(std::true_type{}, std::true_type{}, 0);
        for (int t = 1; t + 1 < ntiles; ++t) tile(std::false_type{}, std::true_type{}, t);
        tile(std::false_type{}, std::false_type{}, ntiles - 1);
    }
    bstore_rows<DP>(O, a.out, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, half, 1.0f / l);
    if (qvalid && half == 0) a.lse_out[((int64_t)n * a.heads + h) * a.Tq + qi] = m + logf(l);
}

// ---- dQ --------------------------------------------------------------------------------------------------------------------------------
// LDS as the forward.  Prologue: delta[q] = sum_d dO.O (also written out for the dK/dV kernel).  Per key tile t:
//     wait V(t)   dP^T = V(t).dO^T        -> DMA V(t+1)
//     wait K(t)   S^T = K(t).Q^T;  p = exp(s - lse), dS = p o (dP o keep - delta);  dQ^T += K(t)^T.dS^T   -> DMA K(t+1)
template <int DP, bool DROP>
__device__ __forceinline__ void battn_bwd_dq_body(const BAttnArgs& a, const int bx, const int h, const int n) {
    using T = BTile<DP>;
    constexpr int NS = DP / 16, NC = DP / 32, PC = T::PC;
    extern __shared__ __attribute__((aligned(16))) char bsmem[];
    char* __restrict__ Ks = bsmem;
    char* __restrict__ Vs = bsmem + T::BYTES;
    float* __restrict__ Mrow = reinterpret_cast<float*>(bsmem + 2 * T::BYTES);
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int q0 = bx * 32, qi = q0 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const int64_t krow_base = (int64_t)n * a.Tk;
    BStream<DP> ks, vs;
    ks.init(a.k + krow_base * a.ldk + col0, Ks, (int)a.ldk, a.Tk, lane);
    vs.init(a.v + krow_base * a.ldv + col0, Vs, (int)a.ldv, a.Tk, lane);
    const int64_t sidx = ((int64_t)n * a.heads + h) * a.Tq + qi;

    // Prologue (one memory round trip, see the forward): DMA tiles, the three register fragments, the small loads last
    vs.issue(0);
    ks.issue(0);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 Qr[NS], Gr[NS];
    float dl;
    {
        bf16x8 Cr[NS];
        bload_rowfrag<DP>(Gr, a.dctx, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
        bload_rowfrag<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
        bload_rowfrag<DP>(Cr, a.ctx, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
        __builtin_amdgcn_sched_barrier(0);
        bfill_row(Mrow, a.mask ? a.mask + krow_base : nullptr, a.Tk, ntiles * 32, 0.f, -INFINITY, lane);
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)Cr[s][e] * (float)Gr[s][e];
        dl = b_halves_sum(acc);          // delta = sum_d dO.O of this lane's query (each half-wave holds half of the head dimension)
        if (qvalid && half == 0) a.delta_out[sidx] = dl;
    }
    const float lse = qvalid ? a.lse[sidx] : INFINITY;          // a query past the end: p = exp(-inf) = 0
    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    f32x16 dQ[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[c][r] = 0.f;
    const uint64_t* const kblock = DROP ? a.keep + (((int64_t)n * a.heads + h) * ((a.Tq + 31) >> 5) + bx) * ntiles * 16 : nullptr;
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

    auto tile = [&](auto FIRST_T, auto MORE_T, const int t) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(FIRST_T)::value, more = decltype(MORE_T)::value;
        const int j0 = t * 32;
        u32x16 ma, mb;          // this block's 16 keep masks (2 dwords each), by scalar loads: they do not touch the counted vmcnt queue
        if (DROP) {
            const uint64_t* kp = scalar_ptr(kblock + (int64_t)t * 16);          // (wave-uniform; s_nop: a scalar the VALU has just written needs wait states before SMEM reads it)
            asm volatile("s_nop 4\n\ts_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(ma), "=&s"(mb) : "s"(kp) : "memory");
        }
        // V(t); K(t) may still be on its way (first tile: the prologue's last loads have been consumed, so everything has landed)
        if (FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else b_wait<PC>();
        f32x16 dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) dP[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) dP = MFMA_B(T::kc(Vs, l31, half, s), Gr[s], dP);
        b_reads_done();
        if (more) vs.issue(j0 + 32);
        // K(t) (and Q in the first tile); V(t+1) may stay in flight
        if (more) b_wait<PC>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) S = MFMA_B(T::kc(Ks, l31, half, s), Qr[s], S);
        float dS[16];
        if (DROP) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ma), "+s"(mb) : : "memory");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mk = blds4(Mrow + j0 + 8 * g + 4 * half);
            const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float p = __expf(bscore(S[r], a.scale, mkv[u]) - lse);
                float dp = dP[r];
                if (DROP) {          // dp = keep ? dp / (1 - p_drop) : 0 -- the mask of register r is the select operand itself
                    const uint64_t m = r < 8 ? ((uint64_t)ma[2 * (r & 7) + 1] << 32) | ma[2 * (r & 7)] : ((uint64_t)mb[2 * (r & 7) + 1] << 32) | mb[2 * (r & 7)];
                    const float scaled = dp * ik;
                    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(dp) : "v"(scaled), "s"(m));
                }
                dS[r] = p * (dp - dl);
            }
        }
        const bf16x8 Sb[2] = {bpack8(dS), bpack8(dS + 8)};
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int c = 0; c < NC; ++c) dQ[c] = MFMA_B(T::tr(Ks, lane, mm, c), Sb[mm], dQ[c]);
        b_reads_done();
        if (more) ks.issue(j0 + 32);
    };
    if (ntiles == 1) {
        tile(std::true_type{}, std::false_type{}, 0);
    } else {
        tile(std::true_type{}, std::true_type{}, 0);
        for (int t = 1; t + 1 < ntiles; ++t) tile(std::false_type{}, std::true_type{}, t);
        tile(std::false_type{}, std::false_type{}, ntiles - 1);
    }
    bstore_rows<DP>(dQ, a.dq, a.lddq, (int64_t)n * a.Tq, qi, a.Tq, col0, half, a.scale);
}

// ---- dK / dV ----------------------------------------------------------------------------------------------------------------------------
// The wave owns 32 keys and loops over query tiles.  LDS: [Q tile][dO tile][lse row, +inf past Tq][delta row].  Per query tile t:
//     wait Q(t)    S = Q(t).K^T        (rows = queries of the tile down the registers, column = this lane's key)
//     wait dO(t)   dP = dO(t).V^T;     p = exp(s - lse), dS = p o (dP o keep - delta)
//     dK^T += Q(t)^T.dS   -> DMA Q(t+1)          dV^T += dO(t)^T.(P o keep)   -> DMA dO(t+1)
// PART: 0 = dK and dV in one pass (one wave per SIMD at d = 128: 128 accumulator registers beside the K and V fragments); 1 = dV only, 2 = dK only.
// The two halves recompute S (and share nothing else: dV needs P, dK needs dS = P o (dP - delta)), 40 instead of 32 matrix instructions per
// tile, but each fits 256 registers, so TWO waves share a SIMD and one wave's ~200-390 vector instructions per tile run under the other's matrix
// instructions -- the one-pass kernel serialises them (4100 cycles per tile against 1024 of matrix time, profiles/round6 notes in LABNOTES).
template <int DP, bool DROP, int PART = 0>
__device__ __forceinline__ void battn_bwd_dkv_body(const BAttnArgs& a, const int bx, const int h, const int n) {
    constexpr bool WANT_V = PART != 2, WANT_K = PART != 1;
    using T = BTile<DP>;
    constexpr int NS = DP / 16, NC = DP / 32, PC = T::PC;
    extern __shared__ __attribute__((aligned(16))) char bsmem[];
    char* __restrict__ Qs = bsmem;
    char* __restrict__ Gs = bsmem + T::BYTES;
    const int nqt = (a.Tq + 31) >> 5;
    float* __restrict__ Lrow = reinterpret_cast<float*>(bsmem + 2 * T::BYTES);
    float* __restrict__ Drow = Lrow + nqt * 32;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int k0 = bx * 32, kj = k0 + l31;
    const bool kvalid = kj < a.Tk;
    const int col0 = h * a.d;
    const int64_t qrow_base = (int64_t)n * a.Tq;
    BStream<DP> qs, gs;
    qs.init(a.q + qrow_base * a.ldq + col0, Qs, (int)a.ldq, a.Tq, lane);
    gs.init(a.dctx + qrow_base * a.ldo + col0, Gs, (int)a.ldo, a.Tq, lane);
    const int64_t srow = ((int64_t)n * a.heads + h) * a.Tq;

    // Prologue (one memory round trip, see the forward): DMA tiles, the two register fragments, the small loads last
    qs.issue(0);
    gs.issue(0);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 Kr[NS], Vr[WANT_K ? NS : 1];
    bload_rowfrag<DP>(Kr, a.k, a.ldk, (int64_t)n * a.Tk, kj, a.Tk, col0, half);
    if constexpr (WANT_K) bload_rowfrag<DP>(Vr, a.v, a.ldv, (int64_t)n * a.Tk, kj, a.Tk, col0, half);
    __builtin_amdgcn_sched_barrier(0);
    bfill_row(Lrow, a.lse + srow, a.Tq, nqt * 32, 0.f, INFINITY, lane);
    if constexpr (WANT_K) bfill_row(Drow, a.delta + srow, a.Tq, nqt * 32, 0.f, 0.f, lane);
    const float mk = kvalid ? (a.mask ? a.mask[(int64_t)n * a.Tk + kj] : 0.f) : -INFINITY;
    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    f32x16 accV[WANT_V ? NC : 1], accK[WANT_K ? NC : 1];
#pragma unroll
    for (int c = 0; c < (WANT_V ? NC : 1); ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) accV[c][r] = 0.f;
#pragma unroll
    for (int c = 0; c < (WANT_K ? NC : 1); ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) accK[c][r] = 0.f;

    // keep bits (see the header): this lane's key is column c = l31 of the block; in the forward's layout that key sat in register rf of half hf
    const int kc = l31, hf = (kc >> 2) & 1, rf = (kc & 3) + 4 * (kc >> 3);
    const int nkt = (a.Tk + 31) >> 5;
    const uint64_t* const kcol = DROP ? a.keep + ((((int64_t)n * a.heads + h) * nqt) * nkt + bx) * 16 + rf : nullptr;

    auto tile = [&](auto MORE_T, const int t) __attribute__((always_inline)) {
        constexpr bool more = decltype(MORE_T)::value;
        const int i0 = t * 32;
        uint32_t kw = 0;          // bit (r & 3) + 8 (r >> 2) of kw = keep decision of query row bkrow(r, half) of this tile for this lane's key
        if (DROP) {
            const uint64_t mk64 = kcol[(int64_t)t * nkt * 16];
            kw = (uint32_t)(hf ? (mk64 >> 32) : mk64) >> (4 * half);
        }
        b_wait<PC>();           // Q(t); dO(t) may still be on its way (first tile: the prologue has drained the queue)
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) S = MFMA_B(T::kc(Qs, l31, half, s), Kr[s], S);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // dO(t)
        f32x16 dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) dP[r] = 0.f;
        if constexpr (WANT_K) {
#pragma unroll
            for (int s = 0; s < NS; ++s) dP = MFMA_B(T::kc(Gs, l31, half, s), Vr[s], dP);
        }
        float Pk[16], dS[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 ls = blds4(Lrow + i0 + 8 * g + 4 * half);
            float4 ds = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (WANT_K) ds = blds4(Drow + i0 + 8 * g + 4 * half);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dsv[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float p = __expf(bscore(S[r], a.scale, mk) - lsv[u]);
                float pk = p, dp = dP[r];
                if (DROP) {
                    const bool keep = (kw & (1u << ((r & 3) + 8 * (r >> 2)))) != 0;
                    pk = keep ? p * ik : 0.f;
                    dp = keep ? dp * ik : 0.f;
                }
                Pk[r] = pk;
                dS[r] = p * (dp - dsv[u]);
            }
        }
        if constexpr (WANT_K) {
            const bf16x8 Sb[2] = {bpack8(dS), bpack8(dS + 8)};
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int c = 0; c < NC; ++c) accK[c] = MFMA_B(T::tr(Qs, lane, mm, c), Sb[mm], accK[c]);          // dK^T += Q^T . dS
        }
        b_reads_done();
        if (more) qs.issue(i0 + 32);
        if constexpr (WANT_V) {
            const bf16x8 Pb[2] = {bpack8(Pk), bpack8(Pk + 8)};
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int c = 0; c < NC; ++c) accV[c] = MFMA_B(T::tr(Gs, lane, mm, c), Pb[mm], accV[c]);          // dV^T += dO^T . (P o keep)
        }
        b_reads_done();
        if (more) gs.issue(i0 + 32);
    };
    for (int t = 0; t + 1 < nqt; ++t) tile(std::true_type{}, t);
    tile(std::false_type{}, nqt - 1);
    if constexpr (WANT_V) bstore_rows<DP>(accV, a.dv, a.lddv, (int64_t)n * a.Tk, kj, a.Tk, col0, half, 1.0f);
    if constexpr (WANT_K) bstore_rows<DP>(accK, a.dk, a.lddk, (int64_t)n * a.Tk, kj, a.Tk, col0, half, a.scale);
}

// ---- kernel entry points: one launch covers ONE problem or the TWO directions of BertBiAttention (see attention.hip) ---------------------
#define YT_BATTN_DECODE(BODY_CALL)                                                                              \
    const int raw = blockIdx.x;                                                                                 \
    const int which = raw < b.nb0 ? 0 : 1;                                                                      \
    const int bid = which ? xcd_remap(raw - b.nb0, (int)gridDim.x - b.nb0) : xcd_remap(raw, b.nb0);             \
    const int gx = which ? b.gx1 : b.gx0;                                                                       \
    const BAttnArgs& a = b.p[which];                                                                            \
    const int bx = bid % gx, h = (bid / gx) % a.heads, n = bid / (gx * a.heads);                                \
    BODY_CALL

template <int DP, bool DROP>
__global__ __launch_bounds__(64, 2) void battn_fwd_kernel(const BAttnLaunch b) { YT_BATTN_DECODE((battn_fwd_body<DP, DROP>(a, bx, h, n))); }
template <int DP, bool DROP>
__global__ __launch_bounds__(64, 2) void battn_bwd_dq_kernel(const BAttnLaunch b) { YT_BATTN_DECODE((battn_bwd_dq_body<DP, DROP>(a, bx, h, n))); }
#ifndef BATTN_DKV_WPS
#define BATTN_DKV_WPS 1          // d = 128: 512 registers and no spills beat two spilling waves per SIMD (img self bwd 2485 -> 2393 us, co pair bwd 1875 -> 1723 us at cfg 5)
#endif
template <int DP, bool DROP>
__global__ __launch_bounds__(64, DP == 128 ? BATTN_DKV_WPS : 2) void battn_bwd_dkv_kernel(const BAttnLaunch b) { YT_BATTN_DECODE((battn_bwd_dkv_body<DP, DROP>(a, bx, h, n))); }
// d = 128 in two passes, two waves per SIMD each (run-time option ATTN_DKV_SPLIT)
template <int DP, bool DROP>
__global__ __launch_bounds__(64, 2) void battn_bwd_dv_kernel(const BAttnLaunch b) { YT_BATTN_DECODE((battn_bwd_dkv_body<DP, DROP, 1>(a, bx, h, n))); }
template <int DP, bool DROP>
__global__ __launch_bounds__(64, 2) void battn_bwd_dk_kernel(const BAttnLaunch b) { YT_BATTN_DECODE((battn_bwd_dkv_body<DP, DROP, 2>(a, bx, h, n))); }
#undef YT_BATTN_DECODE

static int bcheck(const char* who, const BAttnArgs& a) {
    YT_REQUIRE(a.q && a.k && a.v, "%s: null q/k/v", who);
    YT_REQUIRE(a.N > 0 && a.heads > 0 && a.Tq > 0 && a.Tk > 0, "%s: empty problem", who);
    YT_REQUIRE(a.d == 64 || a.d == 128, "%s: head dim %d unsupported (64 or 128)", who, a.d);
    YT_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0, "%s: leading dimensions must be multiples of 8", who);
    YT_REQUIRE((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) & 15) == 0, "%s: q/k/v must be 16-byte aligned", who);
    YT_REQUIRE(a.p_drop >= 0.f && a.p_drop < 1.f, "%s: p_drop out of range", who);
    YT_REQUIRE(!(a.p_drop > 0.f) || a.rng, "%s: dropout needs rng state", who);
    YT_REQUIRE(a.Tk <= 8192 && a.Tq <= 8192, "%s: sequence too long for the LDS-resident mask / lse rows", who);
    YT_REQUIRE((int64_t)a.N * std::max(a.Tq, a.Tk) * std::max(std::max(a.ldq, a.ldk), std::max(a.ldv, a.ldo)) < (1ll << 32),
               "%s: tensor too large for 32-bit element offsets", who);
    return 0;
}

#define YT_BLAUNCH(KERNEL, LDS)                                                                                        \
    do {                                                                                                              \
        if (a0.d == 128) {                                                                                            \
            if (drop) hipLaunchKernelGGL((KERNEL<128, true>), dim3((unsigned)total), dim3(64), LDS, s, b);            \
            else hipLaunchKernelGGL((KERNEL<128, false>), dim3((unsigned)total), dim3(64), LDS, s, b);                \
        } else {                                                                                                      \
            if (drop) hipLaunchKernelGGL((KERNEL<64, true>), dim3((unsigned)total), dim3(64), LDS, s, b);             \
            else hipLaunchKernelGGL((KERNEL<64, false>), dim3((unsigned)total), dim3(64), LDS, s, b);                 \
        }                                                                                                             \
    } while (0)

static int blaunch_fwd(BAttnLaunch& b, int np, hipStream_t s) {
    if (np > 1 && b.p[1].Tk > b.p[0].Tk) std::swap(b.p[0], b.p[1]);          // long workgroups first (attention.hip launch_bwd): forward walks key tiles
    const BAttnArgs& a0 = b.p[0];
    bool drop = false;
    int maxTk = 0;
    for (int i = 0; i < np; ++i) {
        if (int rc = bcheck("attn_fwd_bf16", b.p[i])) return rc;
        YT_REQUIRE(b.p[i].out && b.p[i].lse_out && ((uintptr_t)b.p[i].out & 7) == 0, "attn_fwd_bf16: ctx/lse null or misaligned");
        drop = drop || b.p[i].p_drop > 0.f;
        maxTk = std::max(maxTk, b.p[i].Tk);
    }
    b.gx0 = (int)cdiv(b.p[0].Tq, 32);
    b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tq, 32) : 1;
    b.nb0 = b.gx0 * a0.heads * a0.N;
    const int64_t total = (int64_t)b.nb0 + (np > 1 ? (int64_t)b.gx1 * a0.heads * a0.N : 0);
    YT_REQUIRE(total < (1ll << 31), "attn_fwd_bf16: grid too large");
    const size_t lds = (size_t)2 * 32 * a0.d * 2 + (size_t)cdiv(maxTk, 32) * 32 * sizeof(float);
    for (int i = 0; i < np; ++i)
        YT_REQUIRE(!drop || (b.p[i].keep && ((uintptr_t)b.p[i].keep & 127) == 0),
                   "attn_fwd_bf16: a launch with dropout needs a keep buffer (ytvln_attn_keep_bytes, 128-byte aligned) in every problem");
    YT_BLAUNCH(battn_fwd_kernel, lds);
    YT_LAUNCH_CHECK("attn_fwd_bf16");
    return 0;
}

static int blaunch_bwd(BAttnLaunch& b, int np, hipStream_t s) {
    const BAttnArgs& a0 = b.p[0];
    bool drop = false;
    int maxTq = 0, maxTk = 0;
    for (int i = 0; i < np; ++i) {
        BAttnArgs& a = b.p[i];
        if (int rc = bcheck("attn_bwd_bf16", a)) return rc;
        YT_REQUIRE(a.ctx && a.dctx && a.lse && a.delta && a.dq && a.dk && a.dv, "attn_bwd_bf16: null pointer");
        YT_REQUIRE(a.lddq % 4 == 0 && a.lddk % 4 == 0 && a.lddv % 4 == 0, "attn_bwd_bf16: gradient leading dimensions must be multiples of 4");
        YT_REQUIRE((((uintptr_t)a.ctx | (uintptr_t)a.dctx) & 15) == 0 && (((uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv) & 7) == 0,
                   "attn_bwd_bf16: misaligned pointer");
        a.delta_out = const_cast<float*>(a.delta);
        drop = drop || a.p_drop > 0.f;
        maxTq = std::max(maxTq, a.Tq); maxTk = std::max(maxTk, a.Tk);
    }
    for (int i = 0; i < np; ++i)
        YT_REQUIRE(!drop || (b.p[i].keep && ((uintptr_t)b.p[i].keep & 127) == 0),
                   "attn_bwd_bf16: a launch with dropout needs the keep buffers its forward wrote in every problem");
    {
        if (np > 1 && b.p[1].Tk > b.p[0].Tk) std::swap(b.p[0], b.p[1]);      // dQ walks key tiles: the long-key direction first
        b.gx0 = (int)cdiv(b.p[0].Tq, 32);
        b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tq, 32) : 1;
        b.nb0 = b.gx0 * a0.heads * a0.N;
        const int64_t total = (int64_t)b.nb0 + (np > 1 ? (int64_t)b.gx1 * a0.heads * a0.N : 0);
        YT_REQUIRE(total < (1ll << 31), "attn_bwd_bf16: grid too large");
        const size_t lds = (size_t)2 * 32 * a0.d * 2 + (size_t)cdiv(maxTk, 32) * 32 * sizeof(float);
        YT_BLAUNCH(battn_bwd_dq_kernel, lds);
    }
    {
        if (np > 1 && b.p[1].Tq > b.p[0].Tq) std::swap(b.p[0], b.p[1]);      // dK/dV walks query tiles: the long-query direction first
        b.gx0 = (int)cdiv(b.p[0].Tk, 32);
        b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tk, 32) : 1;
        b.nb0 = b.gx0 * a0.heads * a0.N;
        const int64_t total = (int64_t)b.nb0 + (np > 1 ? (int64_t)b.gx1 * a0.heads * a0.N : 0);
        YT_REQUIRE(total < (1ll << 31), "attn_bwd_bf16: grid too large");
        const size_t lds = (size_t)2 * 32 * a0.d * 2 + (size_t)2 * cdiv(maxTq, 32) * 32 * sizeof(float);
        if (a0.d == 128 && opt(OPT_ATTN_DKV_SPLIT)) {
            const size_t lds_v = (size_t)2 * 32 * a0.d * 2 + (size_t)cdiv(maxTq, 32) * 32 * sizeof(float);          // (no delta row in the dV pass)
            if (drop) { hipLaunchKernelGGL((battn_bwd_dv_kernel<128, true>), dim3((unsigned)total), dim3(64), lds_v, s, b);
                        hipLaunchKernelGGL((battn_bwd_dk_kernel<128, true>), dim3((unsigned)total), dim3(64), lds, s, b); }
            else { hipLaunchKernelGGL((battn_bwd_dv_kernel<128, false>), dim3((unsigned)total), dim3(64), lds_v, s, b);
                   hipLaunchKernelGGL((battn_bwd_dk_kernel<128, false>), dim3((unsigned)total), dim3(64), lds, s, b); }
        } else {
            YT_BLAUNCH(battn_bwd_dkv_kernel, lds);
        }
    }
    YT_LAUNCH_CHECK("attn_bwd_bf16");
    return 0;
}
#undef YT_BLAUNCH

}  // namespace ytvln

using namespace ytvln;

static void bfill(BAttnArgs& a, const ytvln_attn_problem& pr, int N, int heads, int d, float scale, const int64_t* rng) {
    a = BAttnArgs{};
    a.q = (const bf16_t*)pr.q; a.k = (const bf16_t*)pr.k; a.v = (const bf16_t*)pr.v; a.mask = pr.mask;
    a.ctx = (const bf16_t*)pr.ctx_in; a.dctx = (const bf16_t*)pr.dctx; a.lse = pr.lse_in; a.delta = pr.delta;
    a.out = (bf16_t*)pr.ctx; a.lse_out = pr.lse; a.dq = (bf16_t*)pr.dq; a.dk = (bf16_t*)pr.dk; a.dv = (bf16_t*)pr.dv;
    a.ldq = pr.ldq; a.ldk = pr.ldk; a.ldv = pr.ldv; a.ldo = pr.ldo; a.lddq = pr.lddq; a.lddk = pr.lddk; a.lddv = pr.lddv;
    a.N = N; a.heads = heads; a.Tq = pr.Tq; a.Tk = pr.Tk; a.d = d; a.scale = scale; a.p_drop = pr.p_drop; a.rng = rng; a.site = pr.site;
    a.keep = (uint64_t*)pr.keep;
}

extern "C" int64_t ytvln_attn_keep_bytes(int N, int heads, int Tq, int Tk) {
    if (N <= 0 || heads <= 0 || Tq <= 0 || Tk <= 0) return 0;
    return (int64_t)N * heads * cdiv(Tq, 32) * cdiv(Tk, 32) * 128;
}

extern "C" int ytvln_attn_fwd_bf16(const ytvln_attn_problem* pa, const ytvln_attn_problem* pb, int N, int heads, int d, float scale,
                                   const int64_t* rng, void* stream) {
    YT_REQUIRE(pa, "attn_fwd_bf16: null problem");
    BAttnLaunch b = {};
    bfill(b.p[0], *pa, N, heads, d, scale, rng);
    if (pb) bfill(b.p[1], *pb, N, heads, d, scale, rng);
    return blaunch_fwd(b, pb ? 2 : 1, as_stream(stream));
}

extern "C" int ytvln_attn_bwd_bf16(const ytvln_attn_problem* pa, const ytvln_attn_problem* pb, int N, int heads, int d, float scale,
                                   const int64_t* rng, void* stream) {
    YT_REQUIRE(pa, "attn_bwd_bf16: null problem");
    BAttnLaunch b = {};
    bfill(b.p[0], *pa, N, heads, d, scale, rng);
    if (pb) bfill(b.p[1], *pb, N, heads, d, scale, rng);
    return blaunch_bwd(b, pb ? 2 : 1, as_stream(stream));
}
