// Fused multi-head attention (forward + backward) on the CDNA4 fp32 matrix cores.
//
// One code path serves BertSelfAttention, BertImageSelfAttention and both directions of BertBiAttention
// (vilbert/vilbert.py:284-311, 413-440, 552-618): queries and keys/values may come from different streams (Tq != Tk) and
// are read in place from the packed projection outputs through (pointer, leading dimension) pairs.
//
// Register-resident flash attention built on the 32x32x2 f32 MFMA fragment layout, so that no score ever leaves the register
// file and the softmax needs one half-wave exchange:
//   * a wave owns 32 queries and computes the TRANSPOSED score tile  S^T[key][query] = K . Q^T : the accumulator layout
//     (column = lane&31 = query, 16 keys down the registers, the other 16 keys in the partner half-wave) makes the reduction
//     over keys a per-lane loop + one __shfl_xor(32);
//   * those registers ARE the A operand of  O[query][dcol] += P[query][key] . V[key][dcol]  (lane = query row, register r = key
//     krow(r, half) = (r&3) + 8*(r>>2) + 4*half; a sum is order-free), and the B operand is read from the V tile with ONE
//     ds_read_b128 per key row: lane l takes the four consecutive columns 4*l .. 4*l+3, which become column l of four
//     accumulators -- a free permutation of the output columns instead of 4 ds_read_b32 of a transposed operand;
//   * the contraction over the head dimension of K . Q^T is split between half-waves, so the K operand is also one ds_read_b128
//     per 4 MFMAs from an XOR-swizzled row;
//   * O keeps the query down the registers: the (rare, lazily applied) online-softmax rescale and the final 1/l fetch the per-query
//     scalar from the lane that owns the query with one ds_bpermute per register.
// Occupancy is what the tiling is shaped for: <= 256 VGPRs (two waves per SIMD), K and V tiles of 32 keys in ONE
// LDS buffer each (32 KB at d = 128) refilled by LDS-DMA in a half-tile hand-over -- the K buffer is refilled under the P.V matmul, the
// V buffer under the next K.Q^T -- so FOUR two-wave workgroups fit a CU; a workgroup takes 32 x nw queries, waves past the end of
// the sequence skip the matrix work (they only help with the DMA), and consecutive workgroups of one (pair, head) share an XCD.
//
// Backward = recompute-based flash backward in two kernels with the same fragment tricks:
//   dq kernel  (workgroup owns queries, loops over key tiles):  dP^T = V . dO^T, S^T = K . Q^T, dS^T -> dQ += dS . K
//   dkv kernel (workgroup owns keys, loops over query tiles; waves work in PAIRS on one 32-key tile so that each stays under 256 VGPRs):
//       wave 0 of the pair: S = Q . K^T -> P (sent to its partner through 4 KB of LDS), dV += P~^T . dO
//       wave 1 of the pair: dP = dO . V^T, dS = P o (dP~ - delta), dK += dS^T . Q
// The head dimension d may be any multiple of 4 up to 128; it is zero-padded to DP in {32, 64, 128}.
#include "common.h"
#include <algorithm>

#ifndef YT_ATTN_SPREAD
#define YT_ATTN_SPREAD 0        // 1: forward one-wave kernel requests its LDS-DMA pieces between the matrix instructions (round-6 experiment: -0.5 % of the family)
#endif
namespace ytvln {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const float* q; const float* k; const float* v; const float* mask;
    const float* ctx; const float* dctx; const float* lse; const float* delta;
    float* out; float* lse_out; float* dq; float* dk; float* dv;
    float* delta_out;  // dQ kernel: also write delta[n,h,q] = sum_c dctx * ctx (it owns the query rows; the dK/dV kernel, launched after it, reads it)
    int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int N, heads, Tq, Tk, d;
    float scale, p_drop;
    const int64_t* rng; int64_t site;
    int dsplit;        // 1: single-tile two-wave workgroups run the d-split form (the launch reserved the 4 KB exchange buffer)
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#ifndef YT_ATTN_BWD_PIPE
#define YT_ATTN_BWD_PIPE 0      // LDS read batch of the backward kernels' matmuls (0 = compiler order; see mma_rows)
#endif
#define RESCALE_THR 12.0f    // e^12 ~ 1.6e5: far inside fp32 range even summed over thousands of keys

__device__ __forceinline__ int krow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// scaled + masked score with the reference's two roundings (scores / sqrt(d), then + mask; vilbert.py:295-297): the
// additive mask is -10000, so an fma here would change fully-masked rows at the 1e-3 level.
__device__ __forceinline__ float score(float s, float scale, float mask) { return __fadd_rn(__fmul_rn(s, scale), mask); }

// ---- LDS tile streaming ------------------------------------------------------------------------------------------------
// A 32 x DP tile lives in LDS as S[row][DP] with the 16-byte granule g of row r stored at position g ^ (r & 7).  Tiles are
// fed by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write); the DMA writes lane-linear 1 KiB pieces, so the swizzle
// is applied to the per-lane SOURCE address.  Rows past the end / columns past d are CLAMPED to valid data (finite garbage): such
// keys carry a -inf mask (p = 0), such queries a +inf lse (p = 0), and garbage columns only reach accumulator columns that are
// never stored.
// Reads: 4 consecutive dk of the lane's own row = one ds_read_b128 (A operand, <= 2-way conflict); DP/32 consecutive columns of a
// row shared by a half-wave = one contiguous 128..512-byte sweep (B operand, conflict-free).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int DP>
struct Tile {
    static constexpr int PIECES = DP / 8;    // 1 KiB pieces per 32-row tile
    // `base` already points at (first row of this (pair, head)'s sequence, first column of the head): uniform per workgroup, so the
    // per-lane part of a source address is a 32-bit element offset (row * ld + column < 2^31 for every shape the entry points accept).
    __device__ static __forceinline__ void issue(float* S, const float* __restrict__ base, int ld, int row0, int nrows, int d, int wave,
                                                 int nw, int lane) {
        for (int p = wave; p < PIECES; p += nw) {          // wave-uniform
            const int off = p * 256 + 4 * lane;
            const int row = off / DP, pos = (off % DP) >> 2;
            const int g = pos ^ (row & 7);
            const int grow = min(row0 + row, nrows - 1), gcol = min(4 * g, d - 4);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (uint32_t)(grow * ld + gcol)), (lds_ptr_t)(S + p * 256), 16, 0, 0);
        }
    }
};

// Per-lane LDS offsets, computed once per kernel: the swizzle only touches the low 3 granule bits, so every fragment read
// becomes (one of a few lane-constant bases) + (compile-time immediate).
struct LaneOff {
    int rows[8];   // rows[u] : float offset of granule (u ^ (l31&7)) of row l31 (this half-wave's share of the contraction) -> mma_rows
    int brow[4];   // brow[u] : float offset of columns (DP/32)*l31 .. of the row whose (row & 7) == u + 4*half, incl. those rows -> mma_regs_rows
};
template <int DP>
__device__ __forceinline__ LaneOff make_lane_off(int l31, int half) {
    LaneOff o;
#pragma unroll
    for (int u = 0; u < 8; ++u) o.rows[u] = l31 * DP + 4 * ((half * (DP / 8) + u) ^ (l31 & 7));   // (for DP = 32 the half bit is swizzled too)
    const int col = (DP / 32) * l31;
#pragma unroll
    for (int u = 0; u < 4; ++u) o.brow[u] = (u + 4 * half) * DP + 4 * ((col >> 2) ^ (u + 4 * half)) + (col & 3);
    return o;
}

// LDS reads in the tile loops go through __restrict__ parameters: a ds_read without alias information makes hipcc emit
// s_waitcnt vmcnt(0) in front of it whenever an LDS-DMA is in flight (it might alias the DMA's destination), which drains the
// prefetch of the next tile in the middle of the current one.
// Before a tile buffer is handed back to the LDS-DMA every LDS read of it must have RETURNED: the reads feed matrix instructions, which hipcc
// may sink -- together with the s_waitcnt lgkmcnt in front of them -- below the next global_load_lds (seen in the ISA of the bf16 kernels,
// attention_bf16.hip: b_reads_done, where a DMA that hits in L1 / L2 overtook queued reads).  A compiler barrier does not order that; a wait does.
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float4 lds4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float2 lds2(const float* __restrict__ p) { return *reinterpret_cast<const float2*>(p); }

// per-lane operand registers: X[row0 + (lane&31)][half*(DP/2) + s], s = 0..DP/2-1 (zero past nrows / past d)
template <int DP>
__device__ __forceinline__ void load_rowfrag(float (&R)[DP / 2], const float* __restrict__ base, int64_t ld, int64_t row_base,
                                             int row, int nrows, int col0, int d, int half) {
    // branch-free: a load under a per-lane condition is followed by a wait at the join, which serialises the DP/8 loads of a fragment (one
    // memory round trip EACH in every kernel's prologue); clamped addresses + a select keep them all in flight together
    const bool rok = row < nrows;
    const float* __restrict__ rp = base + (row_base + (rok ? row : nrows - 1)) * ld + col0;
    float4 v[DP / 8];
#pragma unroll
    for (int s4 = 0; s4 < DP / 2; s4 += 4) v[s4 >> 2] = *reinterpret_cast<const float4*>(rp + min(half * (DP / 2) + s4, d - 4));
#pragma unroll
    for (int s4 = 0; s4 < DP / 2; s4 += 4) {
        const bool ok = rok && half * (DP / 2) + s4 < d;
        R[s4] = ok ? v[s4 >> 2].x : 0.f; R[s4 + 1] = ok ? v[s4 >> 2].y : 0.f;
        R[s4 + 2] = ok ? v[s4 >> 2].z : 0.f; R[s4 + 3] = ok ? v[s4 >> 2].w : 0.f;
    }
}
// The same fragment for d == DP WITHOUT the zeroing: a row past the end repeats the last row.  The zeroing is a use of the loaded values, and
// hipcc puts a use (and the wait for it) right behind its loads -- a kernel that wants the loads of several fragments and tiles in flight
// together while it already computes on the first (attn_bwd_dq_w1_body) cannot afford that.  Callers make repeated rows harmless themselves
// (a query past the end has lse = +inf, so p = 0; its outputs are not stored; lanes = queries never mix in the matrix products).
template <int DP>
__device__ __forceinline__ void load_rowfrag_raw(float (&R)[DP / 2], const float* __restrict__ base, int64_t ld, int64_t row_base,
                                                 int row, int nrows, int col0, int half) {
    const float* __restrict__ rp = base + (row_base + min(row, nrows - 1)) * ld + col0 + half * (DP / 2);
#pragma unroll
    for (int s4 = 0; s4 < DP / 2; s4 += 4) {
        const float4 v = *reinterpret_cast<const float4*>(rp + s4);
        R[s4] = v.x; R[s4 + 1] = v.y; R[s4 + 2] = v.z; R[s4 + 3] = v.w;
    }
}

// acc (32x32)[tile row][lane's own row] = Xs-tile (A: rows = lane&31 of the LDS tile, contraction split by half) . Rfrag^T (B: registers)
// `fill(slot)`: called once per slot of four (mma_rows: 16 slots at DP = 128) / DP/32 (mma_regs_rows: 16 slots) matrix instructions, between them,
// order pinned -- the one-wave kernels put the LDS-DMA pieces of the NEXT tile there (YT_ATTN_SPREAD) instead of issuing a whole tile in a burst
// with the matrix pipe idle.
struct NoFill { __device__ __forceinline__ void operator()(int) const {} };
template <int DP, int PIPE = 0, class F = NoFill>
__device__ __forceinline__ f32x16 mma_rows(const float* __restrict__ Xs, const float (&R)[DP / 2], const LaneOff& lo, F fill = F{}) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (PIPE > 0) {
        // LDS reads in batches of PIPE granules, batch b+1 issued between the matrix instructions of batch b: one exposed LDS latency per
        // call instead of one per pair of reads (hipcc on its own waits for every pair right where it issues it).  Costs ~2 x PIPE live
        // registers more than the compiler's order: 8 in the forward kernel, 4 where the register budget is tight.
        constexpr int NG = DP / 8, HB = NG > PIPE ? PIPE : NG, NB = NG / HB;      // granules per half-wave; batch size; batches
        float4 x[2][HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) x[0][u] = lds4(Xs + lo.rows[u & 7] + (u & ~7) * 4);
        __builtin_amdgcn_sched_barrier(0);          // (the scheduler would sink the reads back to their first use)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int gi = b * HB + u;
                if (b + 1 < NB) x[(b + 1) & 1][u] = lds4(Xs + lo.rows[(gi + HB) & 7] + ((gi + HB) & ~7) * 4);
                acc = MFMA(x[b & 1][u].x, R[4 * gi], acc);
                acc = MFMA(x[b & 1][u].y, R[4 * gi + 1], acc);
                acc = MFMA(x[b & 1][u].z, R[4 * gi + 2], acc);
                acc = MFMA(x[b & 1][u].w, R[4 * gi + 3], acc);
                if constexpr (!std::is_same<F, NoFill>::value) { __builtin_amdgcn_sched_barrier(0); fill(gi); __builtin_amdgcn_sched_barrier(0); }
                else if (b + 1 < NB) __builtin_amdgcn_sched_barrier(0);
            }
        }
        return acc;
    } else {
#pragma unroll
        for (int s4 = 0; s4 < DP / 2; s4 += 4) {
            // granule index within the half = s4/4; its low 3 bits are swizzled (lane-constant table), the rest is an immediate
            const float4 x = lds4(Xs + lo.rows[(s4 >> 2) & 7] + ((s4 >> 2) & ~7) * 4);
            acc = MFMA(x.x, R[s4], acc);
            acc = MFMA(x.y, R[s4 + 1], acc);
            acc = MFMA(x.z, R[s4 + 2], acc);
            acc = MFMA(x.w, R[s4 + 3], acc);
        }
        return acc;
    }
}

// acc[j][own row (registers)][column (DP/32)*lane + j] += P (A: own registers, contraction over the 32 tile rows in krow order)
//                                                          . Xs-tile (B: DP/32 consecutive columns of tile row krow(r, half))
// NR = 8: only tile rows 0..15 (krow(r, half), r < 8, are exactly those) -- the last tile of a sequence whose length is 16 mod 32 (the 80-token
// text: 2.5 tiles) holds nothing past row 15 that is not multiplied by an exact zero (p = 0 behind a -inf mask / a +inf lse), so the matmuls
// that CONTRACT over the tile rows drop that half: same bits, half the matrix instructions and LDS reads of the tail.  (The matmuls whose OUTPUT
// rows are the tile rows cannot: a 32x32 instruction produces all 32; see DESIGN section 7 for what a 16x16x4 tail would add and cost.)
template <int DP, int PIPE = 0, class F = NoFill, int NR = 16>
__device__ __forceinline__ void mma_regs_rows(f32x16 (&acc)[DP / 32], const float (&P)[16], const float* __restrict__ Xs, const LaneOff& lo, F fill = F{}) {
    constexpr int NJ = DP / 32;
    static_assert(NR == 16 || NR == 8, "whole tile or its first 16 rows");
    if constexpr (PIPE > 0) {
        // tile rows krow(r, half), r = 0..NR-1, in batches of PIPE reads (see mma_rows)
        constexpr int HB = PIPE > NR ? NR : PIPE, NB = NR / HB;
        auto rd = [&](int r, float (&v)[NJ]) {
            const float* p = Xs + lo.brow[r & 3] + 8 * (r >> 2) * DP;
            if constexpr (NJ == 4) { const float4 t = lds4(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
            else if constexpr (NJ == 2) { const float2 t = lds2(p); v[0] = t.x; v[1] = t.y; }
            else v[0] = *p;
        };
        float v[2][HB][NJ];
#pragma unroll
        for (int r = 0; r < HB; ++r) rd(r, v[0][r]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int r = b * HB + u;
                if (b + 1 < NB) rd(r + HB, v[(b + 1) & 1][u]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = MFMA(P[r], v[b & 1][u][j], acc[j]);
                if constexpr (!std::is_same<F, NoFill>::value) { __builtin_amdgcn_sched_barrier(0); fill(r); __builtin_amdgcn_sched_barrier(0); }
                else if (b + 1 < NB) __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) {         // tile row krow(r, half)
            const float* p = Xs + lo.brow[r & 3] + 8 * (r >> 2) * DP;
            if constexpr (NJ == 4) {
                const float4 v = lds4(p);
                acc[0] = MFMA(P[r], v.x, acc[0]); acc[1] = MFMA(P[r], v.y, acc[1]);
                acc[2] = MFMA(P[r], v.z, acc[2]); acc[3] = MFMA(P[r], v.w, acc[3]);
            } else if constexpr (NJ == 2) {
                const float2 v = lds2(p);
                acc[0] = MFMA(P[r], v.x, acc[0]); acc[1] = MFMA(P[r], v.y, acc[1]);
            } else {
                acc[0] = MFMA(P[r], *p, acc[0]);
            }
        }
    }
}

// acc[j][register r <-> row krow(r, half)][column NJ*l31 + j] -> global rows row0 + krow(r, half) < nrows, columns < d, times mul
template <int DP>
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[DP / 32], float* __restrict__ base, int64_t ld, int64_t row_base, int row0,
                                           int nrows, int col0, int d, int l31, int half, float mul) {
    constexpr int NJ = DP / 32;
    if (NJ * l31 >= d) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + krow(r, half);
        if (row < nrows) {
            float* p = base + (row_base + row) * ld + col0 + NJ * l31;
            if constexpr (NJ == 4) *reinterpret_cast<float4*>(p) = make_float4(acc[0][r] * mul, acc[1][r] * mul, acc[2][r] * mul, acc[3][r] * mul);
            else if constexpr (NJ == 2) *reinterpret_cast<float2*>(p) = make_float2(acc[0][r] * mul, acc[1][r] * mul);
            else *p = acc[0][r] * mul;
        }
    }
}

// DMA pieces issued by this wave have landed in LDS, this wave's ds_writes are done, and every wave of the workgroup got here
#define TILE_WAIT_AND_SYNC()                                            \
    do {                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                                   \
    } while (0)

// ------------------------------------------------------------------------------------------------------------------
// forward.  LDS: [K tile][V tile][mask row, -inf past Tk].  Per key tile t:
//     wait+barrier (K(t) landed, everyone finished P.V(t-1))  -> DMA V(t)   | S^T = K(t).Q^T, softmax
//     wait+barrier (V(t) landed, everyone finished K(t).Q^T)  -> DMA K(t+1) | O += P.V(t)
template <int DP, bool DROP>
__device__ __forceinline__ void attn_fwd_dsplit_body(const AttnArgs& a, const int bx, const int h, const int n);      // below

template <int DP, bool DROP>
__device__ __forceinline__ void attn_fwd_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP, NJ = DP / 32;
    if constexpr (DP == 128) {
        // a two-wave workgroup holding a single query tile shares it between its waves (d-split form above); workgroup-uniform
        if (a.dsplit && blockDim.x == 128 && (bx * 2 + 1) * 32 >= a.Tq) {
            attn_fwd_dsplit_body<DP, DROP>(a, bx, h, n);
            return;
        }
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + TS;
    float* Mrow = smem + 2 * TS;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int q0 = (bx * nw + wave) * 32;          // this wave's queries q0 .. q0+31
    const bool active = q0 < a.Tq;                 // wave-uniform: a wave past the end only helps with the DMA and the barriers
    const int qi = q0 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const int64_t krow_base = (int64_t)n * a.Tk;
    const float* __restrict__ kb = a.k + krow_base * a.ldk + col0;      // this (pair, head)'s K / V rows: uniform bases for the DMA
    const float* __restrict__ vb = a.v + krow_base * a.ldv + col0;
    const int ldk = (int)a.ldk, ldv = (int)a.ldv;
    const LaneOff lo = make_lane_off<DP>(l31, half);

    Tile<DP>::issue(Ks, kb, ldk, 0, a.Tk, a.d, wave, nw, lane);      // K(0) travels while Q and the mask row are fetched

    float Qr[DP / 2];
    load_rowfrag<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, a.d, half);
    for (int j = tid; j < ntiles * 32; j += nthr) Mrow[j] = j < a.Tk ? (a.mask ? a.mask[krow_base + j] : 0.f) : -INFINITY;

    f32x16 O[NJ];
#pragma unroll
    for (int c = 0; c < NJ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    float P[16];

    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    const uint32_t dlo = (uint32_t)((((int64_t)n * a.heads + h) * a.Tq + qi));      // score row id; element = (row id, key)
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 32;
        TILE_WAIT_AND_SYNC();
        Tile<DP>::issue(Vs, vb, ldv, j0, a.Tk, a.d, wave, nw, lane);
        if (active) {
            const f32x16 S = mma_rows<DP, 8>(Ks, Qr, lo);
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
            // rescaled) when some row's tile maximum exceeds it by more than RESCALE_THR.  Mathematically identical (any
            // reference cancels in O / l); exp arguments stay <= RESCALE_THR.  O holds the query down the registers, so the
            // per-query factor comes from the lane that owns that query (same half-wave) -- 16 ds_bpermute, a few times per kernel.
            if (__any(mt > m + RESCALE_THR)) {
                const float mn = fmaxf(m, mt);
                const float alpha = __expf(m - mn);
                l *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ar = __shfl(alpha, krow(r, half) + 32 * half, 64);
#pragma unroll
                    for (int c = 0; c < NJ; ++c) O[c][r] *= ar;
                }
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
        }
        TILE_WAIT_AND_SYNC();
        if (t + 1 < ntiles) Tile<DP>::issue(Ks, kb, ldk, j0 + 32, a.Tk, a.d, wave, nw, lane);
        if (active) mma_regs_rows<DP, 8>(O, P, Vs, lo);
    }
    if (active) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ir = __shfl(inv, krow(r, half) + 32 * half, 64);
#pragma unroll
            for (int c = 0; c < NJ; ++c) O[c][r] *= ir;
        }
        store_rows<DP>(O, a.out, a.ldo, (int64_t)n * a.Tq, q0, a.Tq, col0, a.d, l31, half, 1.0f);
        if (qvalid && half == 0) a.lse_out[((int64_t)n * a.heads + h) * a.Tq + qi] = m + logf(l);   // lse is reference-independent
    }
}

// ---- forward, "d-split" form for a workgroup whose two waves share ONE query tile -------------------------------------------------
// With two-wave workgroups a sequence of 9 query tiles leaves the fifth workgroup of every (pair, head) with one tile: instead of idling,
// its second wave takes half of the head dimension.  Wave w contracts K.Q^T over columns [w*DP/2, (w+1)*DP/2) (32 instead of 64 matrix
// instructions), the two partial score tiles are added through a 4 KB LDS buffer (wave 1 adds and publishes, so both waves hold bit-identical
// scores and run the same softmax), and wave w accumulates / stores output columns 4*l + 2w, 4*l + 2w + 1 (two of the four accumulators).  The
// workgroup leaves its CU slot in ~60 % of the time of a full one, which takes the partial third round out of a 2240-workgroup launch.
// DP = 128 and fp32 operands only (the launch reserves the exchange buffer exactly when this form can occur).
template <int DP, bool DROP>
__device__ __forceinline__ void attn_fwd_dsplit_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP, QN = DP / 4;         // QN: contraction values per lane (this wave's half of d, split between half-waves)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + TS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int ntiles = (a.Tk + 31) >> 5;
    float* Mrow = smem + 2 * TS;
    float* X = Mrow + ntiles * 32;                   // [4 register groups][64 lanes] x 4 floats: the score exchange
    const int q0 = bx * 2 * 32;
    const int qi = q0 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int64_t krow_base = (int64_t)n * a.Tk;
    const float* __restrict__ kb = a.k + krow_base * a.ldk + col0;
    const float* __restrict__ vb = a.v + krow_base * a.ldv + col0;
    const int ldk = (int)a.ldk, ldv = (int)a.ldv;
    // A operand: granules w*(DP/8) + half*(DP/16) + u, u = 0..DP/16-1 (= 8 for DP = 128: exactly the three swizzled bits)
    int arow[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) arow[u] = l31 * DP + 4 * (w * (DP / 8) + half * (DP / 16)) + 4 * (u ^ (l31 & 7));
    const LaneOff lo = make_lane_off<DP>(l31, half);
    const int bsel = 2 * w;                          // this wave's pair of output columns inside a lane's four

    Tile<DP>::issue(Ks, kb, ldk, 0, a.Tk, a.d, w, 2, lane);

    float Qr[QN];
#pragma unroll
    for (int s4 = 0; s4 < QN; s4 += 4) {
        const int col = w * (DP / 2) + half * QN + s4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qvalid && col < a.d) v = *reinterpret_cast<const float4*>(a.q + ((int64_t)n * a.Tq + qi) * a.ldq + col0 + col);
        Qr[s4] = v.x; Qr[s4 + 1] = v.y; Qr[s4 + 2] = v.z; Qr[s4 + 3] = v.w;
    }
    for (int j = tid; j < ntiles * 32; j += 128) Mrow[j] = j < a.Tk ? (a.mask ? a.mask[krow_base + j] : 0.f) : -INFINITY;

    f32x16 O[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    float P[16];
    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    const uint32_t dlo = (uint32_t)((((int64_t)n * a.heads + h) * a.Tq + qi));
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 32;
        TILE_WAIT_AND_SYNC();
        Tile<DP>::issue(Vs, vb, ldv, j0, a.Tk, a.d, w, 2, lane);
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = lds4(Ks + arow[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                S = MFMA(x[u].x, Qr[4 * u], S); S = MFMA(x[u].y, Qr[4 * u + 1], S);
                S = MFMA(x[u].z, Qr[4 * u + 2], S); S = MFMA(x[u].w, Qr[4 * u + 3], S);
            }
        }
        // partial scores: wave 0 publishes, wave 1 adds (one fixed order) and publishes the sum, wave 0 picks it up
        if (w == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(X + (g * 64 + lane) * 4) = make_float4(S[4 * g], S[4 * g + 1], S[4 * g + 2], S[4 * g + 3]);
        }
        TILE_WAIT_AND_SYNC();
        if (w == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = lds4(X + (g * 64 + lane) * 4);
                S[4 * g] = o.x + S[4 * g]; S[4 * g + 1] = o.y + S[4 * g + 1]; S[4 * g + 2] = o.z + S[4 * g + 2]; S[4 * g + 3] = o.w + S[4 * g + 3];
                *reinterpret_cast<float4*>(X + (g * 64 + lane) * 4) = make_float4(S[4 * g], S[4 * g + 1], S[4 * g + 2], S[4 * g + 3]);
            }
        }
        TILE_WAIT_AND_SYNC();
        if (w == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = lds4(X + (g * 64 + lane) * 4);
                S[4 * g] = o.x; S[4 * g + 1] = o.y; S[4 * g + 2] = o.z; S[4 * g + 3] = o.w;
            }
        }
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
        if (__any(mt > m + RESCALE_THR)) {
            const float mn = fmaxf(m, mt);
            const float alpha = __expf(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ar = __shfl(alpha, krow(r, half) + 32 * half, 64);
                O[0][r] *= ar; O[1][r] *= ar;
            }
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
        TILE_WAIT_AND_SYNC();
        if (t + 1 < ntiles) Tile<DP>::issue(Ks, kb, ldk, j0 + 32, a.Tk, a.d, w, 2, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float2 v = lds2(Vs + lo.brow[r & 3] + 8 * (r >> 2) * DP + bsel);
            O[0] = MFMA(P[r], v.x, O[0]);
            O[1] = MFMA(P[r], v.y, O[1]);
        }
    }
    const float inv = 1.0f / l;
    const int colw = 4 * l31 + bsel;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float ir = __shfl(inv, krow(r, half) + 32 * half, 64);
        const int row = q0 + krow(r, half);
        if (row < a.Tq && colw < a.d)
            *reinterpret_cast<float2*>(a.out + ((int64_t)n * a.Tq + row) * a.ldo + col0 + colw) = make_float2(O[0][r] * ir, O[1][r] * ir);
    }
    if (w == 0 && qvalid && half == 0) a.lse_out[((int64_t)n * a.heads + h) * a.Tq + qi] = m + logf(l);
}

// dQ.  LDS as in the forward.  Per key tile t:
//     wait+barrier (V(t) landed, everyone finished dS.K(t-1)) -> DMA K(t)   | dP^T = V(t).dO^T
//     wait+barrier (K(t) landed, everyone finished V(t).dO^T) -> DMA V(t+1) | S^T = K(t).Q^T, dS, dQ += dS.K(t)
template <int DP, bool DROP>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP, NJ = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + TS;
    float* Mrow = smem + 2 * TS;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int q0 = (bx * nw + wave) * 32;
    const bool active = q0 < a.Tq;
    const int qi = q0 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const int64_t krow_base = (int64_t)n * a.Tk;
    const float* __restrict__ kb = a.k + krow_base * a.ldk + col0;
    const float* __restrict__ vb = a.v + krow_base * a.ldv + col0;
    const int ldk = (int)a.ldk, ldv = (int)a.ldv;
    const LaneOff lo = make_lane_off<DP>(l31, half);

    Tile<DP>::issue(Vs, vb, ldv, 0, a.Tk, a.d, wave, nw, lane);      // V(0) travels while the register fragments are fetched

    float Qr[DP / 2], Gr[DP / 2];
    load_rowfrag<DP>(Gr, a.dctx, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, a.d, half);
    const int64_t sidx = ((int64_t)n * a.heads + h) * a.Tq + qi;
    float dl;
    {
        // delta = sum_c dO[q][c] * O[q][c] for this lane's query (each half-wave holds half of the head dimension); written for the dK/dV
        // kernel, which follows on the stream
        float Cr[DP / 2];          // the O fragment: dead before the accumulators come alive; all three fragments' loads fly together
        load_rowfrag<DP>(Cr, a.ctx, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, a.d, half);
        load_rowfrag<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, a.d, half);
        float acc = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < DP / 2; s4 += 4) acc += (Cr[s4] * Gr[s4] + Cr[s4 + 1] * Gr[s4 + 1]) + (Cr[s4 + 2] * Gr[s4 + 2] + Cr[s4 + 3] * Gr[s4 + 3]);
        dl = acc + __shfl_xor(acc, 32, 64);
        if (active && qvalid && half == 0) a.delta_out[sidx] = dl;
    }
    const float lse = qvalid ? a.lse[sidx] : INFINITY;          // a query past the end: p = exp(-inf) = 0
    for (int j = tid; j < ntiles * 32; j += nthr) Mrow[j] = j < a.Tk ? (a.mask ? a.mask[krow_base + j] : 0.f) : -INFINITY;

    f32x16 dQ[NJ];
#pragma unroll
    for (int c = 0; c < NJ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[c][r] = 0.f;

    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    const uint32_t dlo = (uint32_t)sidx;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    f32x16 dP;
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 32;
        TILE_WAIT_AND_SYNC();
        Tile<DP>::issue(Ks, kb, ldk, j0, a.Tk, a.d, wave, nw, lane);
        if (active) dP = mma_rows<DP, YT_ATTN_BWD_PIPE>(Vs, Gr, lo);
        TILE_WAIT_AND_SYNC();
        if (t + 1 < ntiles) Tile<DP>::issue(Vs, vb, ldv, j0 + 32, a.Tk, a.d, wave, nw, lane);
        if (active) {
            const f32x16 S = mma_rows<DP, YT_ATTN_BWD_PIPE>(Ks, Qr, lo);
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
            mma_regs_rows<DP, YT_ATTN_BWD_PIPE>(dQ, dS, Ks, lo);
        }
    }
    if (active) store_rows<DP>(dQ, a.dq, a.lddq, (int64_t)n * a.Tq, q0, a.Tq, col0, a.d, l31, half, a.scale);
}

// counted wait of the one-wave kernels: at most N vector-memory operations (LDS-DMA pieces, fragment loads) still in flight.  The counts are
// multiples of a tile's DP/8 pieces (= a fragment's DP/8 loads): a literal 16 is right for DP = 128 only.
template <int N>
__device__ __forceinline__ void w1_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifndef YT_ATTN_SHORT_TILE
#define YT_ATTN_SHORT_TILE 1          // 0: measurement build without the 16-row last tile (see mma_regs_rows).  d = 128 only: at d = 64 (text self-attention)
                                      // the same branch measured +2.5 % on a ~26 us kernel (profiles/round6_attn_short_tile_ab.log), at d = 128 -1.1 % on the co-attention pair
#endif
#ifndef W1_PIPE
#define W1_PIPE 8          // LDS fragment reads in flight ahead of the matrix instructions of the one-wave kernels (mma_rows / mma_regs_rows); measured: 4 the same within 1 %, 16 slower (backward 703 -> 738 us: registers)
#endif
// One LDS-DMA stream of the one-wave kernels: 32-row x DP-column tiles of a row-major matrix into a private LDS buffer, DP/8 pieces of 1 KiB
// (256/DP rows each; the Tile<DP> layout: granule g of row r at position g ^ (r & 7)).  Per piece of a FULL tile: one add and one 64-bit add on
// the vector unit, one scalar move for M0 (the per-lane offsets row * ld + granule column are lane constants, the LDS side is address-space-3
// arithmetic -- a generic pointer costs a null check per piece); a tile that crosses the end clamps its rows (four VALU per piece).
// Unpadded heads only (d == DP: no column clamp).
typedef __attribute__((address_space(3))) char lds_char;
template <int DP>
struct W1Stream {
    static constexpr int PIECES = DP / 8, RPP = 256 / DP, LPR = DP / 4, NV = 8 / RPP;      // rows per piece, lanes per row, granule-column variants
    const float* __restrict__ base;
    lds_char* lds;
    int ld, lim, lrow_ld, nrows;
    int gcol[NV], roff[PIECES];
    __device__ __forceinline__ void init(const float* b, float* ldsbuf, int ld_, int nrows_, int lane) {
        const int lr = lane / LPR, pos = lane % LPR;          // this lane's row inside a piece, its 16-byte position inside the row
        base = b; lds = (lds_char*)(lds_ptr_t)ldsbuf; ld = ld_; nrows = nrows_; lim = (nrows_ - 1) * ld_; lrow_ld = lr * ld_;
#pragma unroll
        for (int u = 0; u < NV; ++u) gcol[u] = 4 * (pos ^ ((RPP * u + lr) & 7));
#pragma unroll
        for (int p = 0; p < PIECES; ++p) roff[p] = RPP * p * ld_ + lrow_ld + gcol[p % NV];
    }
    // one piece of the tile at row0 (always the clamping form: four more vector instructions than a full tile's piece, branch-free)
    __device__ __forceinline__ void issue_piece(const int row0, const int p) const {
        const int off = min(row0 * ld + RPP * p * ld + lrow_ld, lim) + gcol[p % NV];
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (uint32_t)off), (lds_ptr_t)(lds + p * 1024), 16, 0, 0);
    }
    __device__ __forceinline__ void issue(const int row0) const {
        const int sbase = row0 * ld;
        if (row0 + 32 <= nrows) {
            static_for<PIECES>([&](auto PT) __attribute__((always_inline)) {
                constexpr int p = decltype(PT)::value;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (uint32_t)(sbase + roff[p])), (lds_ptr_t)(lds + p * 1024), 16, 0, 0);
            });
        } else {
            static_for<PIECES>([&](auto PT) __attribute__((always_inline)) {
                constexpr int p = decltype(PT)::value;
                const int off = min(sbase + RPP * p * ld + lrow_ld, lim) + gcol[p % NV];
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (uint32_t)off), (lds_ptr_t)(lds + p * 1024), 16, 0, 0);
            });
        }
    }
};

// forward, "w1" form: ONE wave per workgroup and per SIMD (unpadded d = 128, fp32 operands, sequences <= 512), written straight down like the
// backward kernels below -- S = K.Q^T, then the softmax, then O += P.V; fragment reads batched inside each matmul (PIPE = 8), a whole tile of
// LDS-DMA issued behind the matmul that frees its buffer, no barrier, the wave's own counted vmcnt as the only waits.  4032 single-tile waves are
// 3.94 rounds of the 1024 wave slots (private 33 KB of LDS -> four per CU); the two-wave form pays for its second wave with two barriers per
// tile and relies on the d-split trick for its half-empty workgroups.  Image self-attention (56 pairs) 262 -> 224 us, co-attention pair
// 181 -> 163 us.  Round 3 also built and measured a software-pipelined one-wave form (softmax of tile t written between the matrix instructions
// of S(t+1), DMA pieces spread over the P.V steps; ISA checked): 246 us -- a wave's VALU time is NOT hidden under its own matrix instructions,
// the times add (LABNOTES.md 5c; profiles/round3_attn_w1_probes.log), so the interleaving only added control overhead and was removed.
// Same arithmetic, same order as attn_fwd_body: the forms agree to rounding (tests/test_attention_forms_gpu.py).
template <int DP, bool DROP>
__device__ __forceinline__ void attn_fwd_w1_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP, NJ = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ Ks = smem;
    float* __restrict__ Vs = smem + TS;
    float* __restrict__ Mrow = smem + 2 * TS;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int q0 = bx * 32, qi = q0 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const int64_t krow_base = (int64_t)n * a.Tk;
    const LaneOff lo = make_lane_off<DP>(l31, half);
    W1Stream<DP> ks, vs;
    ks.init(a.k + krow_base * a.ldk + col0, Ks, (int)a.ldk, a.Tk, lane);
    vs.init(a.v + krow_base * a.ldv + col0, Vs, (int)a.ldv, a.Tk, lane);

    int64_t rng_seed = 0, rng_ctr = 0;          // small loads first, consumed in the first tile (see attn_bwd_dq_w1_body)
    if (DROP) { rng_seed = a.rng[0]; rng_ctr = a.rng[1]; }
    const int mrows = ntiles * 32;
    float mreg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int jj = lane + 64 * u;
        const float mv = a.mask ? a.mask[krow_base + min(jj, a.Tk - 1)] : 0.f;
        mreg[u] = jj < a.Tk ? mv : -INFINITY;
    }
    ks.issue(0);
    vs.issue(0);
    __builtin_amdgcn_sched_barrier(0);          // (the counted waits below assume this issue order: pin it -- ADVICE r3)
    float Qr[DP / 2];          // (a query past the end repeats the last one: its row of O and its lse are not stored)
    load_rowfrag_raw<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
    __builtin_amdgcn_sched_barrier(0);

    f32x16 O[NJ];
#pragma unroll
    for (int c = 0; c < NJ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    const uint32_t dlo = (uint32_t)((((int64_t)n * a.heads + h) * a.Tq + qi));
    if (DROP) { thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }
    auto halves_max = [&](float x) __attribute__((always_inline)) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    };
    auto halves_sum = [&](float x) __attribute__((always_inline)) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    };

    auto tile = [&](auto FIRST_T, auto MORE_T, const int t) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(FIRST_T)::value, more = decltype(MORE_T)::value;
        const int j0 = t * 32;
        // K(t); V(t) (DP/8 pieces) may still be on its way -- in the first tile also Q, which the matmul needs as well
#if YT_ATTN_SPREAD
        // SPREAD: a tile's LDS-DMA pieces are requested between the matrix instructions of the matmul that runs while its buffer is free -- V(t)
        // under S(t) (the V buffer is free since P.V(t-1); V(0) comes from the prologue), K(t+1) under P.V(t) (the K buffer is free since S(t)) --
        // two pieces per slot over the first half of the matmul, instead of 16 pieces in a burst behind the matmul that frees the buffer (16 x
        // ~60 cycles of issue with an idle matrix pipe, twice per tile).  Every wait is vmcnt(0): nothing younger than the tile a matmul needs has
        // been requested when it starts.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // K(t)
        constexpr int PCS = DP / 8;
        auto fill_v = [&](int slot) __attribute__((always_inline)) {
            if (!FIRST && 2 * slot < PCS) { vs.issue_piece(j0, 2 * slot); vs.issue_piece(j0, 2 * slot + 1); }
        };
        const f32x16 S = mma_rows<DP, W1_PIPE>(Ks, Qr, lo, fill_v);
        lds_reads_done();
#else
        if (FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else w1_wait<DP / 8>();
        const f32x16 S = mma_rows<DP, W1_PIPE>(Ks, Qr, lo);
        lds_reads_done();
        if (more) ks.issue(j0 + 32);
#endif
        if constexpr (FIRST) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (lane + 64 * u < mrows) Mrow[lane + 64 * u] = mreg[u];
            if (DROP) key = drop_key_of((uint64_t)rng_seed, (uint64_t)rng_ctr, a.site);
        }
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
        mt = halves_max(mt);
        if (__any(mt > m + RESCALE_THR)) {          // the lazy online-softmax reference move (see attn_fwd_body)
            const float mn = fmaxf(m, mt);
            const float alpha = __expf(m - mn);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ar = __shfl(alpha, krow(r, half) + 32 * half, 64);
#pragma unroll
                for (int c = 0; c < NJ; ++c) O[c][r] *= ar;
            }
        }
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { P[r] = __expf(P[r] - m); ps += P[r]; }
        l += halves_sum(ps);
        if (DROP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t bits = attn_drop_hash((uint32_t)(j0 + krow(r, half)), dlo, key);
                P[r] = bits >= thr ? P[r] * ik : 0.f;
            }
        }
#if YT_ATTN_SPREAD
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // V(t)
        auto fill_k = [&](int slot) __attribute__((always_inline)) {          // K(t+1) into Ks, free since S(t)
            if (more && 2 * slot < PCS) { ks.issue_piece(j0 + 32, 2 * slot); ks.issue_piece(j0 + 32, 2 * slot + 1); }
        };
        mma_regs_rows<DP, W1_PIPE>(O, P, Vs, lo, fill_k);
        lds_reads_done();
#else
        // V(t); K(t+1), if there is one, may still be on its way
        if (more) w1_wait<DP / 8>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (YT_ATTN_SHORT_TILE && DP == 128 && !more && a.Tk - j0 <= 16) mma_regs_rows<DP, W1_PIPE, NoFill, 8>(O, P, Vs, lo);          // (wave-uniform; only the last tile can be short)
        else mma_regs_rows<DP, W1_PIPE>(O, P, Vs, lo);
        lds_reads_done();
        if (more) vs.issue(j0 + 32);
#endif
    };
    if (ntiles == 1) {
        tile(std::true_type{}, std::false_type{}, 0);
    } else {
        tile(std::true_type{}, std::true_type{}, 0);
        for (int t = 1; t + 1 < ntiles; ++t) tile(std::false_type{}, std::true_type{}, t);
        tile(std::false_type{}, std::false_type{}, ntiles - 1);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float ir = __shfl(inv, krow(r, half) + 32 * half, 64);
#pragma unroll
        for (int c = 0; c < NJ; ++c) O[c][r] *= ir;
    }
    store_rows<DP>(O, a.out, a.ldo, (int64_t)n * a.Tq, q0, a.Tq, col0, a.d, l31, half, 1.0f);
    if (qvalid && half == 0) a.lse_out[((int64_t)n * a.heads + h) * a.Tq + qi] = m + logf(l);
}

// dQ, "w1" form: ONE wave per workgroup and per SIMD (d > 64, fp32 operands), as attn_fwd_w1_body.  The two-wave form pays three rounds of
// workgroup slots for the 2.19 a 9-tile sequence needs and leaves its LDS fragment reads unpipelined (256 VGPRs); a lone wave has the registers
// to batch them (mma_rows PIPE = 8), needs no barrier, and 4032 single-tile waves are 3.94 rounds.  Per key tile t:
//     wait V(t)                      dP^T = V(t).dO^T      -> DMA V(t+1)
//     wait K(t)                      S^T = K(t).Q^T, dS;   dQ += dS.K(t)      -> DMA K(t+1)
// (every fetch travels under the matrix work of the other buffer; the waits are the wave's own counted vmcnt: the DP/8 pieces of the other tile may
// stay in flight, except for the last K).  Same arithmetic, same order as attn_bwd_dq_body: bit-identical results.
// What is left on the table is the start of a round: 1024 waves open together and ask for 80 KB each (~10 us of HBM time, 10-15 % of a wave's
// life).  Measured and removed (round 3): touching the successor workgroup's fragment rows from the last tile but one (six LDS-DMA dword loads
// per lane into a dump area) halves the prologue of the waves that guessed their successor right (16k -> 7-10k cycles) and makes the kernel
// 2.5 % SLOWER -- the touched lines (6 MB per XCD and round) do not survive in a 4 MB L2 and are fetched twice
// (profiles/round3_attn_w1_dq_prefetch.log).  The fix that remains is a persistent wave that loads its next fragments into spare registers.
template <int DP, bool DROP>
__device__ __forceinline__ void attn_bwd_dq_w1_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP, NJ = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ Ks = smem;
    float* __restrict__ Vs = smem + TS;
    float* __restrict__ Mrow = smem + 2 * TS;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int q0 = bx * 32, qi = q0 + l31;
    const bool qvalid = qi < a.Tq;
    const int col0 = h * a.d;
    const int ntiles = (a.Tk + 31) >> 5;
    const int64_t krow_base = (int64_t)n * a.Tk;
    const float* __restrict__ kb = a.k + krow_base * a.ldk + col0;
    const float* __restrict__ vb = a.v + krow_base * a.ldv + col0;
    const int ldk = (int)a.ldk, ldv = (int)a.ldv;
    const LaneOff lo = make_lane_off<DP>(l31, half);

    W1Stream<DP> ks, vs;
    ks.init(kb, Ks, ldk, a.Tk, lane);
    vs.init(vb, Vs, ldv, a.Tk, lane);
    auto ktile = [&](int row0) __attribute__((always_inline)) { ks.issue(row0); };
    auto vtile = [&](int row0) __attribute__((always_inline)) { vs.issue(row0); };

    // Prologue, ordered so that the matrix work starts on the first 32 KB: a round of 1024 waves asks for 80 KB each at the same moment
    // (K, V tiles and three register fragments), ~17 us of HBM time during which nothing else runs.  dP^T = V(0).dO^T only needs V(0) and dO;
    // Q, K(0) and the O fragment (for delta) follow in that order and land under it.  (vmcnt retires in order: the waits below count what
    // was issued AFTER the thing waited for.)
    const int64_t sidx = ((int64_t)n * a.heads + h) * a.Tq + qi;
    // small loads first: they are consumed in the first tile, when everything issued after them has long been waited for
    int64_t rng_seed = 0, rng_ctr = 0;
    if (DROP) { rng_seed = a.rng[0]; rng_ctr = a.rng[1]; }
    const int mrows = ntiles * 32;          // <= 512 (launch condition): the mask row rides in eight registers until then
    float mreg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int jj = lane + 64 * u;
        const float mv = a.mask ? a.mask[krow_base + min(jj, a.Tk - 1)] : 0.f;
        mreg[u] = jj < a.Tk ? mv : -INFINITY;
    }
    const float lse = qvalid ? a.lse[sidx] : INFINITY;
    float dl;                                // (this kernel always produces delta itself: launch condition)
    // The counted vmcnt waits of the first tile assume EXACTLY this issue order -- V(0), K(0), dO, Q, O -- so it is pinned: nothing but hipcc's
    // scheduling of the day would otherwise keep a plain global load from moving across an LDS-DMA builtin.
    __builtin_amdgcn_sched_barrier(0);
    vtile(0);
    __builtin_amdgcn_sched_barrier(0);
    float Qr[DP / 2], Gr[DP / 2], Cr[DP / 2];
    // (d == DP: the launch takes this kernel only for unpadded heads -- load_rowfrag_raw; two code paths would meet in register copies, i.e. waits)
    ktile(0);          // (both DMA tiles ahead of the register loads: an LDS read next to an LDS-DMA still in flight makes hipcc wait for everything)
    __builtin_amdgcn_sched_barrier(0);
    load_rowfrag_raw<DP>(Gr, a.dctx, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
    __builtin_amdgcn_sched_barrier(0);
    load_rowfrag_raw<DP>(Qr, a.q, a.ldq, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
    __builtin_amdgcn_sched_barrier(0);
    load_rowfrag_raw<DP>(Cr, a.ctx, a.ldo, (int64_t)n * a.Tq, qi, a.Tq, col0, half);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 dQ[NJ];
#pragma unroll
    for (int c = 0; c < NJ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[c][r] = 0.f;
    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    const uint32_t dlo = (uint32_t)sidx;
    if (DROP) { thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }
    // one key tile; FIRST: tile 0, with the Q fragment (DP/8 loads) and, when this kernel also produces delta, the O fragment (DP/8 more) still in
    // the queue behind K(0)
    auto tile = [&](auto FIRST_T, const int t) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(FIRST_T)::value;
        const int j0 = t * 32;
        const bool more = t + 1 < ntiles;
        // V(t).  Behind it in the queue: K(t) (DP/8 pieces); in the first tile K(0) and dO (needed now as well), then Q and O
        if (FIRST) w1_wait<2 * (DP / 8)>();
        else w1_wait<DP / 8>();
        const f32x16 dP = mma_rows<DP, W1_PIPE>(Vs, Gr, lo);
        lds_reads_done();
        if (more) vtile(j0 + 32);
        // K(t) (and, in the first tile, Q).  Behind them: V(t+1) if there is one, and in the first tile the O fragment
        if (FIRST) { if (more) w1_wait<2 * (DP / 8)>(); else w1_wait<DP / 8>(); }
        else if (more) w1_wait<DP / 8>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x16 S = mma_rows<DP, W1_PIPE>(Ks, Qr, lo);
        if constexpr (FIRST) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (lane + 64 * u < mrows) Mrow[lane + 64 * u] = mreg[u];
            if (DROP) key = drop_key_of((uint64_t)rng_seed, (uint64_t)rng_ctr, a.site);
            {          // delta = sum_c dO.O of this lane's query (see attn_bwd_dq_body); the O fragment has had two matmuls to arrive
                float acc = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < DP / 2; s4 += 4) acc += (Cr[s4] * Gr[s4] + Cr[s4 + 1] * Gr[s4 + 1]) + (Cr[s4 + 2] * Gr[s4 + 2] + Cr[s4 + 3] * Gr[s4 + 3]);
                dl = acc + __shfl_xor(acc, 32, 64);
                if (qvalid && half == 0) a.delta_out[sidx] = dl;
            }
        }
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
        if (YT_ATTN_SHORT_TILE && DP == 128 && a.Tk - j0 <= 16) mma_regs_rows<DP, W1_PIPE, NoFill, 8>(dQ, dS, Ks, lo);          // a 16-row last tile: see mma_regs_rows
        else mma_regs_rows<DP, W1_PIPE>(dQ, dS, Ks, lo);
        lds_reads_done();
        if (more) ktile(j0 + 32);
    };
    tile(std::true_type{}, 0);
    for (int t = 1; t < ntiles; ++t) tile(std::false_type{}, t);
    store_rows<DP>(dQ, a.dq, a.lddq, (int64_t)n * a.Tq, q0, a.Tq, col0, a.d, l31, half, a.scale);
}

// dK / dV.  A PAIR of waves owns 32 keys: wave "S" (role 0) keeps the K fragment and the dV accumulators, wave "D" (role 1) the V
// fragment and the dK accumulators -- 128 of the 256 matrix instructions of a (query tile, key tile) pair each, and each under 256
// VGPRs.  LDS: STAGES x [Q tile][dO tile] | pairs x [P exchange, 4 KB] | [lse row, +inf past Tq][delta row].  Per query tile t:
//     wait+barrier (tile t landed; STAGES = 2: everyone finished tile t-1 -> DMA tile t+1 into the other stage)
//     S-wave: S = Q.K^T -> P -> exchange buffer            D-wave: dP = dO.V^T
//     barrier
//     S-wave: dV += (P o keep)^T . dO                      D-wave: dS = P o (dP o keep - delta), dK += dS^T . Q
//     (STAGES = 1: barrier, DMA tile t+1)
template <int DP, bool DROP, int STAGES>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP, NJ = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int pair = wave >> 1, role = wave & 1, npairs = nw >> 1;
    const int nqt = (a.Tq + 31) >> 5;
    float* Xp = smem + STAGES * 2 * TS + pair * 1024;
    float* Lrow = smem + STAGES * 2 * TS + npairs * 1024;
    float* Drow = Lrow + nqt * 32;
    const int l31 = lane & 31, half = lane >> 5;
    const int k0 = (bx * npairs + pair) * 32;      // this pair's keys k0 .. k0+31
    const bool active = k0 < a.Tk;                 // wave-uniform, the same for both waves of a pair
    const int kj = k0 + l31;
    const bool kvalid = kj < a.Tk;
    const int col0 = h * a.d;
    const int64_t qrow_base = (int64_t)n * a.Tq;
    const LaneOff lo = make_lane_off<DP>(l31, half);

    const float* __restrict__ qb = a.q + qrow_base * a.ldq + col0;      // this (pair, head)'s Q / dO rows: uniform bases for the DMA
    const float* __restrict__ gb = a.dctx + qrow_base * a.ldo + col0;
    const int ldq = (int)a.ldq, ldo = (int)a.ldo;
    auto issue = [&](int t) {
        float* st = smem + (STAGES == 2 ? (t & 1) : 0) * 2 * TS;
        Tile<DP>::issue(st, qb, ldq, t * 32, a.Tq, a.d, wave, nw, lane);
        Tile<DP>::issue(st + TS, gb, ldo, t * 32, a.Tq, a.d, wave, nw, lane);
    };
    issue(0);                                   // first Q/dO tile travels while the register fragment and the lse/delta rows are fetched

    float Fr[DP / 2];                           // K fragment (S-wave) or V fragment (D-wave)
    if (role == 0) load_rowfrag<DP>(Fr, a.k, a.ldk, (int64_t)n * a.Tk, kj, a.Tk, col0, a.d, half);
    else load_rowfrag<DP>(Fr, a.v, a.ldv, (int64_t)n * a.Tk, kj, a.Tk, col0, a.d, half);
    const float mk = kvalid ? (a.mask ? a.mask[(int64_t)n * a.Tk + kj] : 0.f) : -INFINITY;
    const int64_t srow = ((int64_t)n * a.heads + h) * a.Tq;
    for (int j = tid; j < nqt * 32; j += nthr) {
        Lrow[j] = j < a.Tq ? a.lse[srow + j] : INFINITY;
        Drow[j] = j < a.Tq ? a.delta[srow + j] : 0.f;
    }

    f32x16 acc[NJ];                             // dV (S-wave) or dK (D-wave): [key (registers)][column NJ*l31 + j]
#pragma unroll
    for (int c = 0; c < NJ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    float W[16];                                // S-wave: P (then P o keep); D-wave: dP (then dS)
    for (int t = 0; t < nqt; ++t) {
        TILE_WAIT_AND_SYNC();
        if (STAGES == 2 && t + 1 < nqt) issue(t + 1);
        const float* Qs = smem + (STAGES == 2 ? (t & 1) : 0) * 2 * TS;
        const float* Gs = Qs + TS;
        const int i0 = t * 32;
        if (active) {
            if (role == 0) {
                // S[query][key]: rows = queries of the tile (krow order down the registers), column = this lane's key
                const f32x16 S = mma_rows<DP, YT_ATTN_BWD_PIPE>(Qs, Fr, lo);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 ls = lds4(Lrow + i0 + 8 * g + 4 * half);
                    W[4 * g] = __expf(score(S[4 * g], a.scale, mk) - ls.x); W[4 * g + 1] = __expf(score(S[4 * g + 1], a.scale, mk) - ls.y);
                    W[4 * g + 2] = __expf(score(S[4 * g + 2], a.scale, mk) - ls.z); W[4 * g + 3] = __expf(score(S[4 * g + 3], a.scale, mk) - ls.w);
                    *reinterpret_cast<float4*>(Xp + (g * 64 + lane) * 4) = make_float4(W[4 * g], W[4 * g + 1], W[4 * g + 2], W[4 * g + 3]);
                }
            } else {
                const f32x16 dP = mma_rows<DP, YT_ATTN_BWD_PIPE>(Gs, Fr, lo);
#pragma unroll
                for (int r = 0; r < 16; ++r) W[r] = dP[r];
            }
        }
        TILE_WAIT_AND_SYNC();                   // P has been written (lgkmcnt) by the S-wave of every pair
        if (active) {
            if (role == 0) {
                if (DROP) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        W[r] = attn_drop_hash((uint32_t)kj, (uint32_t)(srow + i0 + krow(r, half)), key) >= thr ? W[r] * ik : 0.f;
                }
                mma_regs_rows<DP, YT_ATTN_BWD_PIPE>(acc, W, Gs, lo);          // dV += P~^T . dO
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 pv = lds4(Xp + (g * 64 + lane) * 4);
                    const float4 ds = lds4(Drow + i0 + 8 * g + 4 * half);
                    const float pvv[4] = {pv.x, pv.y, pv.z, pv.w}, dsv[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = 4 * g + u;
                        float dp = W[r];
                        if (DROP) dp = attn_drop_hash((uint32_t)kj, (uint32_t)(srow + i0 + krow(r, half)), key) >= thr ? dp * ik : 0.f;
                        W[r] = pvv[u] * (dp - dsv[u]);
                    }
                }
                mma_regs_rows<DP, YT_ATTN_BWD_PIPE>(acc, W, Qs, lo);          // dK += dS^T . Q
            }
        }
        if (STAGES == 1 && t + 1 < nqt) {
            __builtin_amdgcn_s_barrier();       // everyone finished tile t: its buffers may be refilled
            issue(t + 1);
        }
    }
    if (active) {
        if (role == 0) store_rows<DP>(acc, a.dv, a.lddv, (int64_t)n * a.Tk, k0, a.Tk, col0, a.d, l31, half, 1.0f);
        else store_rows<DP>(acc, a.dk, a.lddk, (int64_t)n * a.Tk, k0, a.Tk, col0, a.d, l31, half, a.scale);
    }
}

// dK / dV, "w1" form: ONE wave per workgroup and per SIMD owns 32 keys and does all 256 matrix instructions of a (query tile, key tile) pair
// (unpadded d = 128, fp32 operands; ~340 of the 512 registers a lone wave may use).  The wave-pair form above splits the pair between two
// waves that meet at two barriers per tile and pass P through LDS, and each of them hashes the dropout mask; here P stays in registers, the
// mask is hashed once, the LDS fragment reads are batched (PIPE = 8) and nothing waits for anybody.  Private LDS: [Q tile][dO tile][lse row]
// [delta row] (35 KB: four per CU).  Per query tile t:
//     wait Q(t)    S = Q.K^T                       wait dO(t)    dP = dO.V^T         P, keep, dS
//     dK += dS^T.Q   -> DMA Q(t+1)  (travels under the next matmul)      dV += (P o keep)^T.dO   -> DMA dO(t+1)  (travels under the next S)
// Same arithmetic, same order as attn_bwd_dkv_body: bit-identical results.
template <int DP, bool DROP>
__device__ __forceinline__ void attn_bwd_dkv_w1_body(const AttnArgs& a, const int bx, const int h, const int n) {
    constexpr int TS = 32 * DP, NJ = DP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ Qs = smem;
    float* __restrict__ Gs = smem + TS;
    const int nqt = (a.Tq + 31) >> 5;
    float* __restrict__ Lrow = smem + 2 * TS;
    float* __restrict__ Drow = Lrow + nqt * 32;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int k0 = bx * 32, kj = k0 + l31;
    const bool kvalid = kj < a.Tk;
    const int col0 = h * a.d;
    const int64_t qrow_base = (int64_t)n * a.Tq;
    const LaneOff lo = make_lane_off<DP>(l31, half);
    const float* __restrict__ qb = a.q + qrow_base * a.ldq + col0;
    const float* __restrict__ gb = a.dctx + qrow_base * a.ldo + col0;
    const int ldq = (int)a.ldq, ldo = (int)a.ldo;

    W1Stream<DP> qs, gs;
    qs.init(qb, Qs, ldq, a.Tq, lane);
    gs.init(gb, Gs, ldo, a.Tq, lane);
    auto qtile = [&](int row0) __attribute__((always_inline)) { qs.issue(row0); };
    auto gtile = [&](int row0) __attribute__((always_inline)) { gs.issue(row0); };
    qtile(0);
    gtile(0);
    float Kr[DP / 2], Vr[DP / 2];          // (a key past the end repeats the last one: its mask is -inf, so p = 0, and its rows are not stored)
    load_rowfrag_raw<DP>(Kr, a.k, a.ldk, (int64_t)n * a.Tk, kj, a.Tk, col0, half);
    load_rowfrag_raw<DP>(Vr, a.v, a.ldv, (int64_t)n * a.Tk, kj, a.Tk, col0, half);
    const float mk = kvalid ? (a.mask ? a.mask[(int64_t)n * a.Tk + kj] : 0.f) : -INFINITY;
    const int64_t srow = ((int64_t)n * a.heads + h) * a.Tq;
    {          // lse / delta rows of up to 512 queries (launch condition), +inf / 0 past the end
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = lane + 64 * u;
            if (j < nqt * 32) {
                Lrow[j] = j < a.Tq ? a.lse[srow + j] : INFINITY;
                Drow[j] = j < a.Tq ? a.delta[srow + j] : 0.f;
            }
        }
    }
    f32x16 accV[NJ], accK[NJ];
#pragma unroll
    for (int c = 0; c < NJ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accV[c][r] = 0.f; accK[c][r] = 0.f; }
    DropKey key = {0, 0, 0, 0};
    uint32_t thr = 0; float ik = 1.f;
    if (DROP) { key = make_drop_key(a.rng, a.site); thr = drop_threshold(a.p_drop); ik = 1.0f / (1.0f - a.p_drop); }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // (the last tile is peeled: no conditional DMA issue inside the loop -- control flow there costs register shuffles of both accumulators)
    auto tile = [&](auto MORE_T, const int t) __attribute__((always_inline)) {
        constexpr bool more = decltype(MORE_T)::value;
        const int i0 = t * 32;
        w1_wait<DP / 8>();           // Q(t); dO(t) may still be on its way
        const f32x16 S = mma_rows<DP, W1_PIPE>(Qs, Kr, lo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // dO(t)
        const f32x16 dP = mma_rows<DP, W1_PIPE>(Gs, Vr, lo);
        float Pk[16], dS[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 ls = lds4(Lrow + i0 + 8 * g + 4 * half);
            const float4 ds = lds4(Drow + i0 + 8 * g + 4 * half);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dsv[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                const float p = __expf(score(S[r], a.scale, mk) - lsv[u]);
                float pk = p, dp = dP[r];
                if (DROP) {
                    const bool keep = attn_drop_hash((uint32_t)kj, (uint32_t)(srow + i0 + krow(r, half)), key) >= thr;
                    pk = keep ? p * ik : 0.f;
                    dp = keep ? dp * ik : 0.f;
                }
                Pk[r] = pk;
                dS[r] = p * (dp - dsv[u]);
            }
        }
        const bool short_tile = YT_ATTN_SHORT_TILE && DP == 128 && !more && a.Tq - i0 <= 16;          // a 16-row last tile: see mma_regs_rows
        if (short_tile) mma_regs_rows<DP, W1_PIPE, NoFill, 8>(accK, dS, Qs, lo);
        else mma_regs_rows<DP, W1_PIPE>(accK, dS, Qs, lo);          // dK += dS^T . Q
        lds_reads_done();
        if (more) qtile(i0 + 32);
        if (short_tile) mma_regs_rows<DP, W1_PIPE, NoFill, 8>(accV, Pk, Gs, lo);
        else mma_regs_rows<DP, W1_PIPE>(accV, Pk, Gs, lo);          // dV += (P o keep)^T . dO
        lds_reads_done();
        if (more) gtile(i0 + 32);
    };
    for (int t = 0; t + 1 < nqt; ++t) tile(std::true_type{}, t);
    tile(std::false_type{}, nqt - 1);
    store_rows<DP>(accV, a.dv, a.lddv, (int64_t)n * a.Tk, k0, a.Tk, col0, a.d, l31, half, 1.0f);
    store_rows<DP>(accK, a.dk, a.lddk, (int64_t)n * a.Tk, k0, a.Tk, col0, a.d, l31, half, a.scale);
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

// ---- kernel entry points: one launch covers ONE problem or the TWO directions of BertBiAttention (1-D grid: the first nb0
// workgroups belong to problem 0).  The linear workgroup id is remapped so that consecutive ids -- the workgroups of one (pair, head),
// which stream the same K/V (or Q/dO) rows -- run on the same XCD and share its L2; with two problems in the launch the hardware
// dispatcher fills the slots that one direction's last partial round would leave idle with the other direction's workgroups.
struct AttnLaunch { AttnArgs p[2]; int nb0, gx0, gx1; };

#define YT_ATTN_DECODE(BODY_CALL)                                                                              \
    const int raw = blockIdx.x;                                                                                \
    const int which = raw < b.nb0 ? 0 : 1;                                                                     \
    /* remapped inside the problem: every XCD gets a contiguous share of BOTH directions (their workgroups differ 3x in length) */ \
    const int bid = which ? xcd_remap(raw - b.nb0, (int)gridDim.x - b.nb0) : xcd_remap(raw, b.nb0);            \
    const int gx = which ? b.gx1 : b.gx0;                                                                      \
    const AttnArgs& a = b.p[which];                                                                            \
    const int bx = bid % gx, h = (bid / gx) % a.heads, n = bid / (gx * a.heads);                               \
    BODY_CALL

template <int DP, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnLaunch b) { YT_ATTN_DECODE((attn_fwd_body<DP, DROP>(a, bx, h, n))); }
template <int DP, bool DROP>
__global__ __launch_bounds__(64) void attn_fwd_w1_kernel(const AttnLaunch b) { YT_ATTN_DECODE((attn_fwd_w1_body<DP, DROP>(a, bx, h, n))); }
template <int DP, bool DROP>
__global__ __launch_bounds__(64) void attn_bwd_dq_w1_kernel(const AttnLaunch b) { YT_ATTN_DECODE((attn_bwd_dq_w1_body<DP, DROP>(a, bx, h, n))); }
template <int DP, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnLaunch b) { YT_ATTN_DECODE((attn_bwd_dq_body<DP, DROP>(a, bx, h, n))); }
template <int DP, bool DROP>
__global__ __launch_bounds__(64) void attn_bwd_dkv_w1_kernel(const AttnLaunch b) { YT_ATTN_DECODE((attn_bwd_dkv_w1_body<DP, DROP>(a, bx, h, n))); }
template <int DP, bool DROP, int STAGES>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnLaunch b) { YT_ATTN_DECODE((attn_bwd_dkv_body<DP, DROP, STAGES>(a, bx, h, n))); }
#undef YT_ATTN_DECODE

// Waves per workgroup of the forward / dQ kernels (32 queries each).  d > 64 (256-VGPR kernels, two waves per SIMD, 33 KB of LDS): two-wave
// workgroups -- four of them fill a CU's eight wave slots, and a sequence of 9 query tiles costs one idle wave in five workgroups.  Smaller
// heads run three waves per SIMD, so the shape with the fewest idle waves wins (ties: more waves share a staged tile).
static int pick_waves(int T, int d) {
    const int tiles = (int)cdiv(T, 32);
    if (d > 64) return tiles == 1 ? 1 : 2;
    int best = 1, best_idle = 1 << 30;
    for (int nw = 4; nw >= 1; --nw) {
        const int idle = (int)cdiv(tiles, nw) * nw - tiles;
        if (idle < best_idle) { best_idle = idle; best = nw; }
    }
    return best;
}
// Pairs of waves per workgroup of the dK/dV kernel (32 keys per pair) and the number of Q/dO tile stages.
static int pick_pairs(int T, int d) {
    const int tiles = (int)cdiv(T, 32);
    if (tiles == 1) return 1;
    return d > 64 ? 1 : 2;
}
static int pick_stages(int d, int npairs) { return (d > 64 && npairs == 1) ? 1 : 2; }

static int check_common(const char* who, const AttnArgs& a) {
    YT_REQUIRE(a.q && a.k && a.v, "%s: null q/k/v", who);
    YT_REQUIRE(a.N > 0 && a.heads > 0 && a.Tq > 0 && a.Tk > 0, "%s: empty problem", who);
    YT_REQUIRE(a.d > 0 && a.d % 4 == 0 && a.d <= 128, "%s: head dim %d unsupported (multiple of 4, <= 128)", who, a.d);
    YT_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0, "%s: leading dimensions must be multiples of 4", who);
    YT_REQUIRE(((uintptr_t)a.q & 15) == 0 && ((uintptr_t)a.k & 15) == 0 && ((uintptr_t)a.v & 15) == 0, "%s: q/k/v must be 16-byte aligned", who);
    YT_REQUIRE(a.p_drop >= 0.f && a.p_drop < 1.f, "%s: p_drop out of range", who);
    YT_REQUIRE(!(a.p_drop > 0.f) || a.rng, "%s: dropout needs rng state", who);
    YT_REQUIRE(a.Tk <= 8192 && a.Tq <= 8192, "%s: sequence too long for the LDS-resident mask / lse rows", who);
    return 0;
}

static int dp_of(int d) { return d <= 32 ? 32 : d <= 64 ? 64 : 128; }

// dynamic LDS (bytes)
static size_t lds_fwd(int dp, int Tk) { return (size_t)(2 * 32 * dp + (int)cdiv(Tk, 32) * 32) * sizeof(float); }
static size_t lds_dkv(int dp, int stages, int npairs, int Tq) {
    return (size_t)(stages * 2 * 32 * dp + npairs * 1024 + 2 * (int)cdiv(Tq, 32) * 32) * sizeof(float);
}

#define YT_DISPATCH_DROP(KERNEL, DPV, drop_, ...)                                      \
    do {                                                                               \
        if (drop_) hipLaunchKernelGGL((KERNEL<DPV, true>), __VA_ARGS__);               \
        else hipLaunchKernelGGL((KERNEL<DPV, false>), __VA_ARGS__);                    \
    } while (0)
#define YT_DISPATCH(KERNEL, dp_, drop_, ...)                                           \
    do {                                                                               \
        if ((dp_) == 32) YT_DISPATCH_DROP(KERNEL, 32, drop_, __VA_ARGS__);             \
        else if ((dp_) == 64) YT_DISPATCH_DROP(KERNEL, 64, drop_, __VA_ARGS__);        \
        else YT_DISPATCH_DROP(KERNEL, 128, drop_, __VA_ARGS__);                        \
    } while (0)
#define YT_DKV_DROP(ST, DPV, drop_, ...)                                               \
    do {                                                                               \
        if (drop_) hipLaunchKernelGGL((attn_bwd_dkv_kernel<DPV, true, ST>), __VA_ARGS__);   \
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DPV, false, ST>), __VA_ARGS__);        \
    } while (0)
#define YT_DKV_DP(ST, dp_, drop_, ...)                                                 \
    do {                                                                               \
        if ((dp_) == 32) YT_DKV_DROP(ST, 32, drop_, __VA_ARGS__);                      \
        else if ((dp_) == 64) YT_DKV_DROP(ST, 64, drop_, __VA_ARGS__);                 \
        else YT_DKV_DROP(ST, 128, drop_, __VA_ARGS__);                                 \
    } while (0)

}  // namespace ytvln

using namespace ytvln;

// ---- launches over one or two problems of equal (N, heads, d) ------------------------------------------------------------------
static int launch_fwd(AttnLaunch& b, int np, hipStream_t s) {
    if (np > 1 && b.p[1].Tk > b.p[0].Tk) std::swap(b.p[0], b.p[1]);          // forward workgroups walk key tiles: the long-key direction first (see launch_bwd)
    const AttnArgs& a0 = b.p[0];
    const int dp = dp_of(a0.d);
    bool drop = false;
    int maxTq = 0, maxTk = 0;
    for (int i = 0; i < np; ++i) {
        if (int rc = check_common("attn_fwd", b.p[i])) return rc;
        YT_REQUIRE(b.p[i].out && b.p[i].lse_out && ((uintptr_t)b.p[i].out & 15) == 0, "attn_fwd: ctx/lse null or misaligned");
        drop = drop || b.p[i].p_drop > 0.f;
        maxTq = std::max(maxTq, b.p[i].Tq); maxTk = std::max(maxTk, b.p[i].Tk);
    }
    // one wave per workgroup and per SIMD (attn_fwd_w1_body); option ATTN_W1 bit 0 clear: the two-wave form everywhere
    const bool w1_dim = a0.d == 128 || a0.d == 64;             // unpadded heads the one-wave kernels are instantiated for
    if ((opt(OPT_ATTN_W1) & 1) && w1_dim && maxTk <= 512) {
        b.gx0 = (int)cdiv(b.p[0].Tq, 32);
        b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tq, 32) : 1;
        b.nb0 = b.gx0 * a0.heads * a0.N;
        const int64_t total = (int64_t)b.nb0 + (np > 1 ? (int64_t)b.gx1 * a0.heads * a0.N : 0);
        YT_REQUIRE(total < (1ll << 31), "attn_fwd: grid too large");
        for (int i = 0; i < np; ++i) b.p[i].dsplit = 0;
#define YT_W1(KERNEL, LDS)                                                                                             \
    do {                                                                                                              \
        if (a0.d == 128) {                                                                                            \
            if (drop) hipLaunchKernelGGL((KERNEL<128, true>), dim3((unsigned)total), dim3(64), LDS, s, b);            \
            else hipLaunchKernelGGL((KERNEL<128, false>), dim3((unsigned)total), dim3(64), LDS, s, b);                \
        } else {                                                                                                      \
            if (drop) hipLaunchKernelGGL((KERNEL<64, true>), dim3((unsigned)total), dim3(64), LDS, s, b);             \
            else hipLaunchKernelGGL((KERNEL<64, false>), dim3((unsigned)total), dim3(64), LDS, s, b);                 \
        }                                                                                                             \
    } while (0)
        YT_W1(attn_fwd_w1_kernel, lds_fwd(dp, maxTk));
        YT_LAUNCH_CHECK("attn_fwd (w1)");
        return 0;
    }
    const int nw = pick_waves(maxTq, a0.d);
    b.gx0 = (int)cdiv(b.p[0].Tq, 32 * nw);
    b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tq, 32 * nw) : 1;
    b.nb0 = b.gx0 * a0.heads * a0.N;
    const int64_t total = (int64_t)b.nb0 + (np > 1 ? (int64_t)b.gx1 * a0.heads * a0.N : 0);
    YT_REQUIRE(total < (1ll << 31), "attn_fwd: grid too large");
    const bool dsplit = opt(OPT_ATTN_DSPLIT) && dp == 128 && nw == 2;
    for (int i = 0; i < np; ++i) b.p[i].dsplit = dsplit;
    const size_t lds_bytes = lds_fwd(dp, maxTk) + (dsplit ? 4096 : 0);
    YT_DISPATCH(attn_fwd_kernel, dp, drop, dim3((unsigned)total), dim3(64 * nw), lds_bytes, s, b);
    YT_LAUNCH_CHECK("attn_fwd");
    return 0;
}

static int launch_bwd(AttnLaunch& b, int np, hipStream_t s) {
    const AttnArgs& a0 = b.p[0];
    const int dp = dp_of(a0.d);
    bool drop = false;
    int maxTq = 0, maxTk = 0;
    for (int i = 0; i < np; ++i) {
        const AttnArgs& a = b.p[i];
        if (int rc = check_common("attn_bwd", a)) return rc;
        YT_REQUIRE(a.ctx && a.dctx && a.lse && a.delta && a.dq && a.dk && a.dv, "attn_bwd: null pointer");
        YT_REQUIRE(a.lddq % 4 == 0 && a.lddk % 4 == 0 && a.lddv % 4 == 0, "attn_bwd: gradient leading dimensions must be multiples of 4");
        YT_REQUIRE((((uintptr_t)a.ctx | (uintptr_t)a.dctx | (uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv) & 15) == 0, "attn_bwd: misaligned pointer");
        drop = drop || a.p_drop > 0.f;
        maxTq = std::max(maxTq, a.Tq); maxTk = std::max(maxTk, a.Tk);
    }
    // delta[n,h,q] = sum_c dctx.ctx is produced by the dQ kernel's prologue (it owns the query rows) and read by the dK/dV kernel that
    // follows it on the stream
    for (int i = 0; i < np; ++i) b.p[i].delta_out = const_cast<float*>(b.p[i].delta);
    if (np > 1 && b.p[1].Tk > b.p[0].Tk) std::swap(b.p[0], b.p[1]);          // dQ workgroups walk key tiles: the long-key direction first
    {
        // one wave per workgroup and per SIMD (attn_bwd_dq_w1_body); option ATTN_W1 bit 1 clear: the two-wave form
        const bool w1 = (opt(OPT_ATTN_W1) & 2) && (a0.d == 128 || a0.d == 64) && maxTk <= 512;      // (unpadded heads, mask row in registers)
        const int nw = w1 ? 1 : pick_waves(maxTq, a0.d);
        b.gx0 = (int)cdiv(b.p[0].Tq, 32 * nw);
        b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tq, 32 * nw) : 1;
        b.nb0 = b.gx0 * a0.heads * a0.N;
        const int64_t total = (int64_t)b.nb0 + (np > 1 ? (int64_t)b.gx1 * a0.heads * a0.N : 0);
        YT_REQUIRE(total < (1ll << 31), "attn_bwd: grid too large");
        if (w1) YT_W1(attn_bwd_dq_w1_kernel, lds_fwd(dp, maxTk));
        else YT_DISPATCH(attn_bwd_dq_kernel, dp, drop, dim3((unsigned)total), dim3(64 * nw), lds_fwd(dp, maxTk), s, b);
    }
    // one wave per workgroup and per SIMD (attn_bwd_dkv_w1_body; option ATTN_W1 bit 2 clear: always the wave-pair form): unpadded fp32 heads, lse /
    // delta rows staged by one wave, and at least two rounds of the 1024 wave slots (a 1.3-round launch -- 3 key tiles x 448 heads -- pays for 2:
    // there the pair form, whose workgroups are half as long, loses less; option ATTN_W1_DKV_ANY = 1 takes the one-wave form regardless)
    // Two directions in one launch: a dK/dV workgroup owns key tiles and walks the QUERY tiles, so the direction with the longer query sequence
    // has the longer workgroups; those go first (the grid is handed out in order: long ones last would leave a tail of a few hundred long
    // workgroups on a mostly idle chip -- 288-query x 80-key workgroups behind 80 x 288 ones: ~30 instead of ~24 tile times).  The forward / dQ
    // launches walk KEY tiles; BertBiAttention already passes its long-key direction first.
    if (np > 1 && b.p[1].Tq > b.p[0].Tq) std::swap(b.p[0], b.p[1]);
    const int64_t w1_waves = (cdiv(b.p[0].Tk, 32) + (np > 1 ? cdiv(b.p[1].Tk, 32) : 0)) * a0.heads * a0.N;
    const int64_t w1_slots = a0.d == 128 ? 1024 : 2048;          // (d = 64: 17 KB of LDS and < 256 registers per wave -> two per SIMD)
    const bool w1_fill = w1_waves * 100 >= cdiv(w1_waves, w1_slots) * w1_slots * 85;      // the last round at least ~85 % useful overall
    if ((opt(OPT_ATTN_W1) & 4) && (a0.d == 128 || a0.d == 64) && maxTq <= 512 && (w1_fill || opt(OPT_ATTN_W1_DKV_ANY))) {
        b.gx0 = (int)cdiv(b.p[0].Tk, 32);
        b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tk, 32) : 1;
        b.nb0 = b.gx0 * a0.heads * a0.N;
        const int64_t total = w1_waves;
        YT_REQUIRE(total < (1ll << 31), "attn_bwd: grid too large");
        const size_t lds = (size_t)(2 * 32 * dp + 2 * (int)cdiv(maxTq, 32) * 32) * sizeof(float);
        YT_W1(attn_bwd_dkv_w1_kernel, lds);
#undef YT_W1
    } else {
        const int npairs = pick_pairs(maxTk, a0.d), stages = pick_stages(a0.d, npairs);
        b.gx0 = (int)cdiv(b.p[0].Tk, 32 * npairs);
        b.gx1 = np > 1 ? (int)cdiv(b.p[1].Tk, 32 * npairs) : 1;
        b.nb0 = b.gx0 * a0.heads * a0.N;
        const int64_t total = (int64_t)b.nb0 + (np > 1 ? (int64_t)b.gx1 * a0.heads * a0.N : 0);
        YT_REQUIRE(total < (1ll << 31), "attn_bwd: grid too large");
        const size_t lds = lds_dkv(dp, stages, npairs, maxTq);
        if (stages == 1) YT_DKV_DP(1, dp, drop, dim3((unsigned)total), dim3(128 * npairs), lds, s, b);
        else YT_DKV_DP(2, dp, drop, dim3((unsigned)total), dim3(128 * npairs), lds, s, b);
    }
    YT_LAUNCH_CHECK("attn_bwd");
    return 0;
}

extern "C" int ytvln_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                  const float* mask, float* ctx, int64_t ldo, float* lse, int N, int heads, int Tq, int Tk,
                                  int d, float scale, float p_drop, const int64_t* rng, int64_t site, void* stream) {
    AttnLaunch b = {};
    AttnArgs& a = b.p[0];
    a.q = q; a.k = k; a.v = v; a.mask = mask; a.out = ctx; a.lse_out = lse;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.N = N; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d; a.scale = scale; a.p_drop = p_drop; a.rng = rng; a.site = site;
    return launch_fwd(b, 1, as_stream(stream));
}

extern "C" int ytvln_attn_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                  const float* mask, const float* ctx, const float* dctx, int64_t ldo, const float* lse,
                                  float* delta, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                                  int N, int heads, int Tq, int Tk, int d, float scale, float p_drop, const int64_t* rng,
                                  int64_t site, void* stream) {
    AttnLaunch b = {};
    AttnArgs& a = b.p[0];
    a.q = q; a.k = k; a.v = v; a.mask = mask; a.ctx = ctx; a.dctx = dctx; a.lse = lse; a.delta = delta;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.N = N; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d; a.scale = scale; a.p_drop = p_drop; a.rng = rng; a.site = site;
    return launch_bwd(b, 1, as_stream(stream));
}

// ---- both directions of BertBiAttention in one launch -----------------------------------------------------------------------
static void fill_args(AttnArgs& a, const ytvln_attn_problem& pr, int N, int heads, int d, float scale, const int64_t* rng) {
    a = AttnArgs{};
    a.q = pr.q; a.k = pr.k; a.v = pr.v; a.mask = pr.mask;
    a.ctx = pr.ctx_in; a.dctx = pr.dctx; a.lse = pr.lse_in; a.delta = pr.delta;
    a.out = pr.ctx; a.lse_out = pr.lse; a.dq = pr.dq; a.dk = pr.dk; a.dv = pr.dv;
    a.ldq = pr.ldq; a.ldk = pr.ldk; a.ldv = pr.ldv; a.ldo = pr.ldo; a.lddq = pr.lddq; a.lddk = pr.lddk; a.lddv = pr.lddv;
    a.N = N; a.heads = heads; a.Tq = pr.Tq; a.Tk = pr.Tk; a.d = d; a.scale = scale; a.p_drop = pr.p_drop; a.rng = rng; a.site = pr.site;
}

extern "C" int ytvln_attn_fwd_pair(const ytvln_attn_problem* pa, const ytvln_attn_problem* pb, int N, int heads, int d, float scale,
                                   const int64_t* rng, void* stream) {
    YT_REQUIRE(pa && pb, "attn_fwd_pair: null problem");
    AttnLaunch b = {};
    fill_args(b.p[0], *pa, N, heads, d, scale, rng);
    fill_args(b.p[1], *pb, N, heads, d, scale, rng);
    return launch_fwd(b, 2, as_stream(stream));
}

extern "C" int ytvln_attn_bwd_pair(const ytvln_attn_problem* pa, const ytvln_attn_problem* pb, int N, int heads, int d, float scale,
                                   const int64_t* rng, void* stream) {
    YT_REQUIRE(pa && pb, "attn_bwd_pair: null problem");
    AttnLaunch b = {};
    fill_args(b.p[0], *pa, N, heads, d, scale, rng);
    fill_args(b.p[1], *pb, N, heads, d, scale, rng);
    return launch_bwd(b, 2, as_stream(stream));
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
