// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, register-prefetched, double-buffered.
//
// Replaces every nn.Linear on the ViLBERT path and its backward (see include/ytvln.h).  fp32-in / fp32-accumulate MFMA
// is bit-for-bit an fmaf chain (MI355X guide section 3), so parity with the reference's fp32 arithmetic holds to rounding order.
//
// Tiling: workgroup = 256 threads = 4 waves (2x2); block tile BM x BN x 32; each wave owns (BM/2) x (BN/2) as a grid of
// 32x32 MFMA tiles.  LDS holds both operands k-major ( S[k][m] ) so a wave's operand fetch is 32 consecutive floats per
// half-wave (conflict-free ds_read_b32): lane l supplies A[m = l&31][k = l>>5] and B[k = l>>5][n = l&31].
// Operands whose K dimension is contiguous in memory (x[M,K], nn.Linear weight [N,K]) are transposed on the LDS write
// (row pitch BM+1 -> conflict-free); operands with M/N contiguous (dY^T, x^T views for the backward GEMMs) are written
// with 16-byte stores (row pitch BM).
// (That paragraph describes gemm_f32_kernel, the generic register-staged kernel.  The production path is gemm_dma_kernel further
//  down: LDS-DMA operand delivery, 64x64 ... 256x256 tiles, fused epilogues, deterministic split-K; the same kernel also exists in
//  the three-bf16-term form of fp32 (X3, YTVLN_GEMM_SPLIT_BF16X3); bf16 operands have their own kernels in gemm_bf16.hip.)
#include "gemm_tiles.h"

namespace ytvln {

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    using LA = TileLoader<BM, A_KC>;
    using LB = TileLoader<BN, B_KC>;
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 MFMA tiles per wave along m / n (2x2 waves)
    constexpr int SA = BK * LA::LD, SB = BK * LB::LD;  // floats per buffer
    __shared__ __attribute__((aligned(16))) float smem[2 * SA + 2 * SB];
    float* As = smem;
    float* Bs = smem + 2 * SA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    const TileCoord tc = decode_tile(blockIdx.x, g.tiles_m, g.tiles_n, g.splits, g.split_map);
    const int m0 = tc.m * BM, n0 = tc.n * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[LA::NV], rb[LB::NV];
    const int kbeg = tc.split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);          // loaders zero-fill past kend
    const int nk = (kend - kbeg + BK - 1) / BK;
    LA::load(ra, g.A, g.lda, g.M, kend, m0, kbeg, g.vecA, tid);
    LB::load(rb, g.B, g.ldb, g.N, kend, n0, kbeg, g.vecB, tid);
    LA::store(ra, As, tid);
    LB::store(rb, Bs, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            LA::load(ra, g.A, g.lda, g.M, kend, m0, kbeg + (kt + 1) * BK, g.vecA, tid);
            LB::load(rb, g.B, g.ldb, g.N, kend, n0, kbeg + (kt + 1) * BK, g.vecB, tid);
        }
        const float* a_s = As + cur * SA + half * LA::LD + wm0 + l31;
        const float* b_s = Bs + cur * SB + half * LB::LD + wn0 + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = a_s[kk * LA::LD + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = b_s[kk * LB::LD + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            LA::store(ra, As + (cur ^ 1) * SA, tid);
            LB::store(rb, Bs + (cur ^ 1) * SB, tid);
        }
        __syncthreads();
    }

    gemm_epilogue<TM, TN>(g, acc, m0 + wm0, n0 + wn0, l31, half, tc.split);
}

// ------------------------------------------------------------------------------------------------------------------
// Fast path: tiles fed by LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS without passing through VGPRs, no ds_write),
// branch-free addressing (out-of-range rows / columns are CLAMPED to valid ones; their products land in accumulator
// rows / columns the epilogue never stores), one barrier per k-tile with the next tile's DMA in flight under the MFMAs.
// LDS images (the DMA writes lane-linear 1 KiB pieces, so the layout is chosen through the per-lane SOURCE address):
//   K-contiguous operand  : S[m][32]   16-byte granule g of row m stored at position g ^ ((m >> 1) & 7): the 16 rows of every
//                           ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} land on 16 distinct slots;
//   M/N-contiguous operand: S[k][BMN]  k-major                                           -> conflict-free ds_read_b32.
// The contraction of one 32-deep tile is split between half-waves (half h owns k in [16h, 16h+16)), so a lane feeds four
// consecutive MFMAs from one 16-byte LDS read.  Requires K % 32 == 0 and 16-byte aligned operands (else: generic kernel).
// KB: k-tile depth (32 or 16), NS: LDS ring depth (NS - 1 k-tiles of DMA in flight under the MFMAs), WPS: waves per SIMD the
// register budget is sized for (= resident workgroups per CU x NW / 4).
//
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// X3 = true ("fp32 by three bf16 terms"): the operands stay fp32 in HBM and LDS; a lane splits the eight consecutive k it owns
// (two 16-byte fragments) EXACTLY into x = hi + mid + lo, each term a bf16 (8 significant bits: hi = the top 16 bits of x,
// mid = the top 16 bits of x - hi, lo = x - hi - mid, which has at most 8 significant bits left), and the product a.b is
// accumulated in fp32 from the six largest of the nine cross terms on v_mfma_f32_32x32x16_bf16:
//     hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid        (dropped: mid.lo + lo.mid + lo.lo <= 2^-23 |a||b|)
// i.e. the error per product is of the order of one fp32 rounding of that product, at 48 instead of 128 MFMA passes per 16 k.
struct Split3 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ uint32_t top_halves(float odd, float even) {      // {bf16 bits of odd, bf16 bits of even}, truncated
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}
__device__ __forceinline__ Split3 split3(const float4 u, const float4 v) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    float r1[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        r1[e] = x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xffff0000u);          // exact: the low 16 bits of the significand
        r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xffff0000u);        // exact: its low 8 bits
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = top_halves(x[2 * e + 1], x[2 * e]);
        m[e] = top_halves(r1[2 * e + 1], r1[2 * e]);
        l[e] = top_halves(r2[2 * e + 1], r2[2 * e]);
    }
    Split3 s;
    s.hi = __builtin_bit_cast(bf16x8, h); s.mid = __builtin_bit_cast(bf16x8, m); s.lo = __builtin_bit_cast(bf16x8, l);
    return s;
}

#ifndef YT_GEMM_PROBE
#define YT_GEMM_PROBE 0
#endif
#if YT_GEMM_PROBE
// measurement build (tools/gemm_f32_probe.py, -DYT_GEMM_PROBE=1): waves 0 and 4 of workgroup 3 of EVERY launch stamp s_memtime per k-tile at
// [top | barrier released | k-group 0 issued | 1 | 2 | 3] for k-tiles 4..13 into this buffer (64 words per wave)
__device__ uint32_t g_f32_probe[128];
#endif
#ifndef YT_GEMM_PRIO
#define YT_GEMM_PRIO 1          // 1 (shipped): the younger half of the workgroup (waves NW/2..) at s_setprio 1 for the whole loop -- the probe shows the older wave of
                                // every SIMD winning the matrix pipe, finishing its k-tile ~4500 cycles early and parking at the barrier while its partner runs
                                // alone; priority for the younger half measured +0.5 % on the GEMM family (profiles/round6_gemm_prio_ab.log); 2 (priority
                                // alternating per k-group) measured -2.5 %; 0 = none
#endif
#ifndef YT_GEMM_DMA_SPREAD
#define YT_GEMM_DMA_SPREAD 1
#endif
template <int BM, int BN, bool A_KC, bool B_KC, int NW, int KB, int NS, int WPS, bool X3 = false, int WN = 2>
__global__ __launch_bounds__(NW * 64, WPS) void gemm_dma_kernel(const GemmArgs g) {
    using TA = DmaTile<BM, A_KC, NW, KB>;
    using TB = DmaTile<BN, B_KC, NW, KB>;
    constexpr int WM = NW / WN;                        // waves along m x WN along n (2, or 8 for the 224-row tile: eight waves of 224x32)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * TM * 32 == BM && WN * TN * 32 == BN, "wave grid does not cover the tile");
    constexpr int SA = BM * KB, SB = BN * KB, STAGE = SA + SB;
    constexpr int NPT = TA::NI + TB::NI;               // DMA instructions per wave per k-tile
    constexpr int NG = KB / 8;                         // 4-deep k-groups per half-wave per k-tile
    __shared__ __attribute__((aligned(16))) float smem[NS * STAGE];    // ONE shared object (see guide: DMA + 2nd object de-pipelines)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const TileCoord tc = decode_tile(blockIdx.x, g.tiles_m, g.tiles_n, g.splits, g.split_map);
    const int m0 = tc.m * BM, n0 = tc.n * BN;
    const int kbeg = tc.split * g.kchunk;
    const int kend = min(g.Kloop, kbeg + g.kchunk);
    const int nk = (kend - kbeg) / KB;
    const bool tail_here = g.ktail && kend == g.Kloop;       // this workgroup's last k-tiles cross K

    const float* pa[TA::NI];
    const float* pb[TB::NI];
#pragma unroll
    for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg, wave, lane, i, 0x7fffffff);
#pragma unroll
    for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg, wave, lane, i, 0x7fffffff);
    const int64_t sa = TA::step(g.lda), sb = TB::step(g.ldb);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Row sums of op(A) over this workgroup's k range (db = sum over rows of dY, riding on dW = dY^T X): every tile column reads the same
    // A panel, so the workgroups of tile column 0 (and there the waves of wave column 0) add up the fragments they feed to the MFMAs.
    const bool do_asum = !A_KC && !X3 && g.asum != nullptr && tc.n == 0 && (wave % WN) == 0;
    float asum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;

    int st_in = 0;      // ring slot the next issue() fills
    auto issue = [&](int kt, bool advance = true) {
        float* As = smem + st_in * STAGE;
        float* Bs = As + SA;
        st_in = (st_in + 1 == NS) ? 0 : st_in + 1;
        if (tail_here && kbeg + (kt + 1) * KB > g.K) {      // rare: K tail -- recompute the sources with clamped k rows
#pragma unroll
            for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg + kt * KB, wave, lane, i, g.K - 1);
#pragma unroll
            for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg + kt * KB, wave, lane, i, g.K - 1);
        }
#pragma unroll
        for (int i = 0; i < TA::NI; ++i) {
            if (TA::own(wave, i)) __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[i], (lds_ptr_t)(As + (wave * TA::NI + i) * 256), 16, 0, 0);
            pa[i] += advance ? sa : 0;
        }
#pragma unroll
        for (int i = 0; i < TB::NI; ++i) {
            if (TB::own(wave, i)) __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[i], (lds_ptr_t)(Bs + (wave * TB::NI + i) * 256), 16, 0, 0);
            pb[i] += advance ? sb : 0;
        }
    };

#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue(t, !YT_GEMM_DMA_SPREAD || X3 || t + 1 < nk);          // (SPREAD re-requests the last tile: the pointers never leave the matrix)
    int st_out = 0;     // ring slot the MFMAs read
#if YT_GEMM_PROBE
    uint32_t ts = 0;
    const bool probing = blockIdx.x == 3 && (wave & 3) == 0 && !X3;
#define YT_STAMP(P) do { if (probing && kt >= 4 && kt < 14) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); ts = lane == (kt - 4) * 6 + (P) ? now_ : ts; } } while (0)
#else
#define YT_STAMP(P) do { } while (0)
#endif
#if YT_GEMM_PRIO == 1
    if (!X3 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    for (int kt = 0; kt < nk; ++kt) {
        YT_STAMP(0);
        // my pieces of tile kt have landed (NS-2 younger tiles may stay in flight); after the barrier everybody's have, and
        // everybody is done reading the slot tile kt+NS-1 is about to overwrite (it held tile kt-1)
        if (NS > 2 && nk - kt - 1 >= NS - 2) wait_vmcnt<(NS - 2) * NPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        YT_STAMP(1);
#if YT_GEMM_DMA_SPREAD
        // SPREAD: the next tile's LDS-DMA pieces are not issued here in one burst -- both waves of every SIMD did that at the same moment, behind
        // the barrier, with the matrix pipe idle (8 pieces x 2 waves x 30-60 cycles of issue: the ~1500 cycles a k-tile measured over its 16384
        // of matrix time) -- but one at a time between the matrix instructions of the first k-groups (native path below; X3 keeps the burst).
        const bool inc = kt + NS - 1 < nk;
        float* const As_in = smem + st_in * STAGE;
        float* const Bs_in = As_in + SA;
        if (X3) { if (inc) issue(kt + NS - 1); }
        else if (inc) {
            st_in = (st_in + 1 == NS) ? 0 : st_in + 1;
            if (tail_here && kbeg + (kt + NS) * KB > g.K) {
#pragma unroll
                for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg + (kt + NS - 1) * KB, wave, lane, i, g.K - 1);
#pragma unroll
                for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg + (kt + NS - 1) * KB, wave, lane, i, g.K - 1);
            }
        }
        // piece p of the incoming tile (A pieces first); past the last tile the same request with a zero pointer step (a slot nobody reads any more):
        // no branch inside the matrix stream
        const int64_t sa_in = kt + NS < nk ? sa : 0, sb_in = kt + NS < nk ? sb : 0;          // (no step past the last tile: the re-request stays inside the matrix)
        auto issue_piece = [&](auto pc) __attribute__((always_inline)) {
            constexpr int pidx = decltype(pc)::value;
            if constexpr (pidx < TA::NI) {
                if (TA::own(wave, pidx)) __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[pidx], (lds_ptr_t)(As_in + (wave * TA::NI + pidx) * 256), 16, 0, 0);
                pa[pidx] += sa_in;
            } else if constexpr (pidx < NPT) {
                constexpr int q = pidx - TA::NI;
                if (TB::own(wave, q)) __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[q], (lds_ptr_t)(Bs_in + (wave * TB::NI + q) * 256), 16, 0, 0);
                pb[q] += sb_in;
            }
        };
#else
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
#endif
        const float* As = smem + st_out * STAGE;
        const float* Bs = As + SA;
        st_out = (st_out + 1 == NS) ? 0 : st_out + 1;
        if constexpr (X3) {
            // Software pipeline inside the k-tile: the split of the NEXT operand fragments (VALU) is written next to the six-MFMA groups
            // of the current ones, so that the matrix pipe and the vector ALU of a SIMD work at the same time (5.5 VALU ops per value).
            constexpr int NSP = NG / 2;                        // 16 k per step: this lane's k-groups 2sp and 2sp+1 (8 consecutive k)
            auto fa = [&](int i, int sp) { return split3(TA::frag(As, wm0, i, l31, half, 2 * sp), TA::frag(As, wm0, i, l31, half, 2 * sp + 1)); };
            auto fb = [&](int j, int sp) { return split3(TB::frag(Bs, wn0, j, l31, half, 2 * sp), TB::frag(Bs, wn0, j, l31, half, 2 * sp + 1)); };
            Split3 a[TM], an[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = fa(i, 0);
            Split3 b = fb(0, 0), bn = b;
#pragma unroll
            for (int sp = 0; sp < NSP; ++sp) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bool more = sp + 1 < NSP;
                    if (j + 1 < TN) bn = fb(j + 1, sp);
                    else if (more) bn = fb(0, sp + 1);
                    if (more) {                                // the next step's A fragments, spread over the middle groups
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            if ((TN >= TM + 2 && j == i + 1) || (TN < TM + 2 && j == TN - 1)) an[i] = fa(i, sp + 1);
                    }
                    // term-major so that consecutive matrix instructions target different accumulators (smallest terms first)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].mid, b.mid, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].lo, b.hi, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b.lo, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].mid, b.hi, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b.mid, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].hi, b.hi, acc[i][j], 0, 0, 0);
                    b = bn;
                }
                if (sp + 1 < NSP) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = an[i];
                }
            }
        } else {
#pragma unroll
        for (int sg = 0; sg < NG; ++sg) {
#if YT_GEMM_PRIO == 2
            if (((sg & 1) != 0) == (wave >= NW / 2)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = TA::frag(As, wm0, i, l31, half, sg);
            if constexpr (!A_KC) {
                if (do_asum) {             // wave-uniform: first tile column, first wave column
#pragma unroll
                    for (int i = 0; i < TM; ++i) asum[i] += (a[i].x + a[i].y) + (a[i].z + a[i].w);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = TB::frag(Bs, wn0, j, l31, half, sg);
#if YT_GEMM_DMA_SPREAD
            // pieces [sg * PPG, (sg + 1) * PPG) of the incoming tile, spread over the first half of the k-groups: one behind each accumulator's four
            // matrix instructions (256 cycles of the pipe per accumulator against 30-60 of issue), order pinned
            constexpr int NGH = NG >= 2 ? NG / 2 : 1, PPG = (NPT + NGH - 1) / NGH;
            static_for<TM * TN>([&](auto ijc) {
                constexpr int ij = decltype(ijc)::value, i = ij / TN, j = ij % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                if constexpr (ij < PPG) {
                    if (sg < NGH) {          // (sg is a constant after unrolling)
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<NGH>([&](auto gc) {
                            constexpr int gsel = decltype(gc)::value;
                            if (sg == gsel) {          // a wave with fewer accumulators than pieces per group issues several behind one accumulator
                                static_for<(PPG + TM * TN - 1) / (TM * TN)>([&](auto rc) {
                                    constexpr int off = ij + decltype(rc)::value * TM * TN;
                                    if constexpr (off < PPG) issue_piece(std::integral_constant<int, gsel * PPG + off>{});
                                });
                            }
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            });
#else
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
#endif
            YT_STAMP(2 + sg);
        }
        }
    }
    if constexpr (!A_KC && !X3) {
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
#if YT_GEMM_PROBE
    if (probing) g_f32_probe[(wave >> 2) * 64 + lane] = ts;
#endif
#if YT_GEMM_DMA_SPREAD
    if (!X3) wait_vmcnt<0>();          // the trailing (repeated) requests have landed before this workgroup's LDS is handed on
#endif
    gemm_epilogue<TM, TN>(g, acc, m0 + wm0, n0 + wn0, l31, half, tc.split);
}

// The three-term (X3) kernels are instantiated in their own translation unit -- gemm_x3.hip includes this file with YT_GEMM_X3_TU
// defined -- so that the two halves of the GEMM code compile in parallel.  128x128 tiles run as 4 waves of 64x64 there (two workgroups
// per CU: 7.3 instead of 11 VALU ops per MFMA, 154 -> 166 TFLOP/s on 16128x1024x1024 forced onto that tile).
void launch_x3(int bm, int bn, const GemmArgs& g, int transA, int transB, unsigned grid, hipStream_t s);
#if defined(YT_GEMM_X3_TU)
template <int BM, int BN, int NW, int WPS>
static void launch_x3_tile(const GemmArgs& g, int transA, int transB, unsigned grid, hipStream_t s) {
    const dim3 gr(grid), blk(NW * 64);
    if (!transA && transB) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, true, true, NW, 32, 2, WPS, true>), gr, blk, 0, s, g);
    else if (!transA && !transB) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, true, false, NW, 32, 2, WPS, true>), gr, blk, 0, s, g);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, false, false, NW, 32, 2, WPS, true>), gr, blk, 0, s, g);
    else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, false, true, NW, 32, 2, WPS, true>), gr, blk, 0, s, g);
}
void launch_x3(int bm, int bn, const GemmArgs& g, int transA, int transB, unsigned grid, hipStream_t s) {
    if (bm == 256 && bn == 256) launch_x3_tile<256, 256, 8, 2>(g, transA, transB, grid, s);     // 8 waves of 64x128, one workgroup per CU
    else if (bm == 128 && bn == 128) launch_x3_tile<128, 128, 4, 2>(g, transA, transB, grid, s);
    else launch_x3_tile<128, 64, 4, 2>(g, transA, transB, grid, s);
}
}  // namespace ytvln
#else       // ---- everything below belongs to the main translation unit -------------------------------------------------------------

#ifndef YT_SPLITK_REDUCE_UNROLL
#define YT_SPLITK_REDUCE_UNROLL 4          // partial tiles loaded per batch (1 = the round 1-5 loop)
#endif
// C = sum_s ws[s] (+ bias) (+ beta*C), fixed summation order -> deterministic.  One thread per 4 consecutive columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t ldc,
                                                            const float* __restrict__ bias, int M, int N, int splits, float beta,
                                                            const float* __restrict__ asum_ws, float* __restrict__ asum) {
    if (asum) {          // row sums of op(A) (GemmArgs::asum): same fixed order over the splits
        for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < M; r += (int64_t)gridDim.x * 256) {
            float acc = asum_ws[r];
            for (int s = 1; s < splits; ++s) acc += asum_ws[(int64_t)s * M + r];
            asum[r] = acc;
        }
    }
    const int n4 = (N + 3) >> 2;
    const int64_t total = (int64_t)M * N, groups = (int64_t)M * n4;
    const bool vec = (N & 3) == 0 && (ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    for (int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * 256) {
        const int row = (int)(gidx / n4), col = (int)(gidx % n4) << 2;
        const int64_t i = (int64_t)row * N + col;
        if (vec) {
            float4 acc = *reinterpret_cast<const float4*>(ws + i);
            // (the partial tiles are loaded YT_SPLITK_REDUCE_UNROLL at a time and added in split order: the same sum, but that many 16-byte loads in flight per lane
            // instead of the one or two a run-time trip count leaves -- 1024 blocks of four waves are otherwise short of the ~40 KB per CU that
            // 5 TB/s at ~2 us of latency needs)
            int s = 1;
            for (; s + YT_SPLITK_REDUCE_UNROLL <= splits; s += YT_SPLITK_REDUCE_UNROLL) {
                float4 v[YT_SPLITK_REDUCE_UNROLL];
#pragma unroll
                for (int u = 0; u < YT_SPLITK_REDUCE_UNROLL; ++u) v[u] = *reinterpret_cast<const float4*>(ws + (int64_t)(s + u) * total + i);
#pragma unroll
                for (int u = 0; u < YT_SPLITK_REDUCE_UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; s < splits; ++s) {
                const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)s * total + i);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (bias) { acc.x += bias[col]; acc.y += bias[col + 1]; acc.z += bias[col + 2]; acc.w += bias[col + 3]; }
            float4* cp = reinterpret_cast<float4*>(C + (int64_t)row * ldc + col);
            if (beta != 0.f) { const float4 o = *cp; acc.x += beta * o.x; acc.y += beta * o.y; acc.z += beta * o.z; acc.w += beta * o.w; }
            *cp = acc;
        } else {
            for (int u = 0; u < 4 && col + u < N; ++u) {
                float acc = ws[i + u];
                for (int s = 1; s < splits; ++s) acc += ws[(int64_t)s * total + i + u];
                if (bias) acc += bias[col + u];
                float* cp = C + (int64_t)row * ldc + col + u;
                if (beta != 0.f) acc += beta * *cp;
                *cp = acc;
            }
        }
    }
}

// Launch plan = (tile shape, split count), chosen by a small cost model fitted to measurements on MI355X:
//  * a CU delivers about the same GEMM throughput with one or two resident 128x128 workgroups (0.46 vs 0.49 TFLOP/s), so what a
//    launch pays for is the number of block "waves" ceil(blocks / 256 CUs), not rounds of the 512 resident slots;
//  * per block: a fixed cost (prologue DMA round trip + epilogue) plus k-tiles at the CU-exclusive rate of the tile shape
//    (128x128 / 8 waves: 2.14 us per 32-deep k-tile; 128x64 and 64x64 / 4 waves: 1.14 and 0.55 us);
//  * split-K (plain epilogues only, 128x128 tiles) adds the workspace round trip and the reduce launch.
//    (Round 3, measured and removed: the reduction INSIDE the GEMM -- partials dumped in fragment order with sc1 stores, an arrival counter
//     per tile, the last workgroup to arrive adds them in split order.  Bit-identical to the separate pass, but one workgroup per output
//     tile moves splits x tile bytes at a latency-bound ~50 GB/s while the separate pass uses the whole chip: 119.3 -> 119.7 ms per step
//     when applied to launches with <= 4 splits, 122.4 with <= 8, 126.2 with all.  profiles/round3_fused_splitk.log.)
struct Plan { int tile; int splits; };      // tile: 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 256x128, 4 = 256x256, 5 = 224x256, 6 = 160x256 (3-6: one workgroup per CU)

static double plan_cost(int M, int N, int K, int tile, int sp, int epilogue = 0, bool x3 = false, bool ta = false) {
    // the two large tiles move fewer operand bytes and LDS fragments per MFMA: ~3 % / ~7 % above the 128x128 rate per CU
    static const int bm[7] = {128, 128, 64, 256, 256, 224, 160}, bn[7] = {128, 64, 64, 128, 256, 256, 256};
    // us per 32-deep k-tile at the CU-exclusive rate: native fp32 MFMA | three-bf16-term form (fitted on 16128x1024x1024: the split's
    // VALU work is shared best by the wide wave tiles -- 128x128 is VALU-bound, 256x256 matrix-bound)
    // (64x64 and 256x128 exist only for the native instruction: the three-term planner sees their native cost)
    // Round 5: refitted after the interior epilogue stopped spilling (gemm_tiles.h epilogue_interior): the 256x256 tile's fixed cost fell from
    // 25 to ~10 us (ramp + first operands ~4, epilogue 4.5, drain) and its k-tile measures 7.5 us in whole-chip launches (16128x{1024,2048,3072}x1024:
    // 251.8 / 497.8 / 744.3 us = 1 / 2 / 3 rounds of ~249); 128x128: 2.06 us per k-tile, ~1 us fixed (266 / 784 / 1612 us on 4 / 12 / 33 rounds).
    // 224x256 (eight waves of 224x32: seven 32x32 blocks each): 7/8 of the 256x256 k-tile; it exists for the 4480-row text shapes
    // (4480 = 20 x 224: 240 whole tiles at N = 3072 instead of 216 of which 12 are half empty)
    // and 160x256 (five blocks per wave; 4480 = 28 x 160: 252 whole tiles at N = 2304)
    static const double tk_f32[7] = {2.06, 1.14, 0.55, 4.4, 7.5, 6.6, 4.75}, tk_x3[7] = {1.61, 0.82, 0.55, 4.16, 4.94, 4.94, 4.94};
    static const double tfix[7] = {1.0, 3.0, 3.0, 6.0, 10.0, 10.0, 9.0};
    // native instruction, M-contiguous A (the weight-gradient layout: both operands k-major, fragments gathered by ds_read_b32): fitted in
    // round 2 on the cfg-2 weight-gradient shapes (tools/r2_gpu37.sh) -- the 128x128 workgroups run at 2.3 us per k-tile there (half the
    // flops per operand byte: ~3.7 TB/s of LDS-DMA at 120 TFLOP/s, the same delivery ceiling the fp32x3 256x256 kernel meets), the 256x256 at 7.3
    static const double tk_f32_ta[7] = {2.3, 1.14, 0.55, 4.16, 7.3, 7.3, 7.3};
    const double* tk = x3 ? tk_x3 : (ta ? tk_f32_ta : tk_f32);
    const int kchunk = (int)cdiv(cdiv(K, sp), BK) * BK;
    const int splits = (int)cdiv(K, kchunk);
    const double blocks = (double)(cdiv(M, bm[tile]) * cdiv(N, bn[tile])) * splits;
    const double ncu = (double)sk_num_cus();          // 256 on MI355X (the plans pinned in tests/test_abi.py assume it)
    const double waves = ceil(blocks / ncu);
    // the 4-wave tiles only reach their rate with 2-3 blocks co-resident on a CU (one block = one wave per SIMD)
    const double need = tile == 1 ? 2.0 : tile == 2 ? 3.0 : 1.0, per_cu = std::max(1.0, blocks / ncu);
    const double occ = per_cu < need ? need / per_cu : 1.0;
    // fused activations read / write a second matrix in the epilogue: dearer for the 256-row tiles (64-128 KB per workgroup, no overlap)
    double tf = tfix[tile] + ((tile >= 3 && epilogue != YTVLN_EPI_NONE) ? 2.0 : 0.0);
    if (x3 && tile == 4 && splits > 1) tf = 10.0;        // raw partial tiles: no fused epilogue to expose
    double t = waves * (tf + (double)(kchunk / BK) * tk[tile] * occ);
    // us: reduce launch + workspace bytes at ~3 TB/s (fitted on the native plans; the three-term plans were fitted with 5 TB/s: their
    // partials are consumed sooner and mostly hit the memory-side cache)
    // (the 160- / 224-row plans split 2-4 ways over a few MB: measured 10 us on 4480x768x3072 x 3 -- the partials never leave the memory-side cache)
    if (splits > 1) t += (tile >= 5 ? 4.0 : 8.0) + (double)(splits + 1) * (double)M * (double)N * 4.0 / (x3 ? 5.0e6 : tile >= 5 ? 8.0e6 : 3.0e6);
    return t;
}

// big_ok: the 256-row tiles are only used with a K-contiguous A on the LDS-DMA path (an M-contiguous A needs four ds_read_b32 per
// fragment and loses with the wide wave tiles: 121 -> 99 TFLOP/s on 30522x768x4480).
static Plan plan_gemm(int M, int N, int K, int epilogue, bool big_ok = false, bool x3 = false, bool ta = false) {
    Plan best = {0, 1};
    double best_t = 1e30;
    const int force_tile = opt(OPT_GEMM_TILE), force_sp = opt(OPT_GEMM_SPLITS);       // experiment knobs (-1: the planner decides)
    for (int tile = 0; tile < 7; ++tile) {
        if (force_tile >= 0 && tile != force_tile) continue;
        if (tile >= 3 && (!big_ok || M < 256)) continue;
        // 224x256: native instruction, K-contiguous A, launches of ONE round (measured: 4480x3072x768 189.7 -> 170.5 us; the 2400-tile LM decoder
        // gains nothing over 2160 tiles of 256x256 -- multi-round launches already overlap their tiles' fixed costs)
        const int64_t ntile = tile >= 5 ? cdiv(M, tile == 5 ? 224 : 160) * cdiv(N, 256) : 0;
        // (M >= 4096: at 2304 rows -- cfg 2 with K = 1 -- the 160-row plans measured 2 % behind the 128x128 ones in the step, 308.7 -> 301.8 rows/s)
        if (tile >= 5 && (x3 || ta || N < 256 || !opt(OPT_GEMM_T224) || (force_tile < 0 && (ntile > sk_num_cus() || M < 4096)))) continue;
        if (ta && tile == 3) continue;          // (256x128 was never measured with an M-contiguous A)
        // split-K: 128x128 always; 256x256 in the three-term form and -- round 2 -- for the native weight-gradient layout (ta):
        // 1024x1024x16128 323 -> 305 us, 2048x1024x16128 551 -> 505, 768x3072x4480 190 -> 178 (16 x 16, 32 x 8, 36 x 7 workgroups)
        // (round 5: the 224- / 160-row tiles too, as long as tiles x splits stays one round -- 4480x768x3072: 84 tiles of 160x256 x 3 splits)
        const int smax = ((tile == 0 || tile >= 5 || (tile == 4 && (x3 || ta))) && epilogue == YTVLN_EPI_NONE) ? (int)std::min<int64_t>(16, K / 256) : 1;
        for (int sp = 1; sp <= std::max(1, smax); ++sp) {
            if (force_sp >= 0 && sp != std::max(1, std::min(force_sp, std::max(1, smax)))) continue;
            // (short contractions lose: 4480x1024x768 x 3 splits of 256 measured 76 us against 72.5.  Long ones -- round 6, profiles/round6_long_k_plans.log:
            // one round of 160- / 224-row tiles beats six splits of 128x128 wherever it exists, 4480x768x30528 (the LM decoder's input gradient)
            // 1915 -> 1501 us, 4480x768x8192 526 -> 418, 4480x1024x8192 639 -> 571 -- so the K <= 4096 limit of round 5 is gone)
            if (tile >= 5 && sp > 1 && force_tile < 0 && (ntile * sp > sk_num_cus() || K / sp < 512)) break;
            const double t = plan_cost(M, N, K, tile, sp, epilogue, x3, ta);
            // near-ties go to the earlier candidate (fewer splits, the well-trodden 128x128 path); the 256-row tiles only need 0.5 %
            if (t < best_t * (tile >= 3 ? 0.995 : 0.98)) { best_t = t; best = {tile, sp}; }
        }
    }
    return best;
}

static int plan_splits(int M, int N, int K, int epilogue) { return plan_gemm(M, N, K, epilogue).splits; }

template <int BM, int BN>
static int launch_tile(GemmArgs& g, int transA, int transB, hipStream_t s) {
    g.tiles_m = (int)cdiv(g.M, BM);
    g.tiles_n = (int)cdiv(g.N, BN);
    g.ntiles = g.tiles_m * g.tiles_n;
    dim3 grid(g.ntiles * g.splits), block(256);
    if (g.fast) {
#define YT_DMA(NW, KB, NS, WPS)                                                                                                        \
    do {                                                                                                                               \
        dim3 blk(NW * 64);                                                                                                             \
        if (!transA && transB) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, true, true, NW, KB, NS, WPS>), grid, blk, 0, s, g);          \
        else if (!transA && !transB) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, true, false, NW, KB, NS, WPS>), grid, blk, 0, s, g);   \
        else if (transA && !transB) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, false, false, NW, KB, NS, WPS>), grid, blk, 0, s, g);   \
        else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, false, true, NW, KB, NS, WPS>), grid, blk, 0, s, g);                           \
    } while (0)
        // the tiles the three-term planner uses also exist in that form (YTVLN_GEMM_SPLIT_BF16X3): their own translation unit
        if (g.x3 && ((BM == 256 && BN == 256) || (BM == 128 && BN == 128) || (BM == 128 && BN == 64))) {
            launch_x3(BM, BN, g, transA, transB, grid.x, s);
            return 0;
        }
        // 128x128 tiles run 8 waves per workgroup (4 per SIMD at 2 workgroups/CU): measured 113 -> 121 TFLOP/s on
        // 16128x1024x1024 and 84 -> 105 on the split-K weight gradients versus 4 waves (barrier coupling across SIMDs).
        // (Measured and rejected, round 1: 256x128 tiles; 16-deep k-tiles with 2/3/4-stage rings (up to 4 workgroups per CU); forcing
        //  the LDS fragment reads one k-group ahead of the MFMAs.  All within +-3 % of this configuration: in the main loop the matrix
        //  cores are ~88 % busy, the rest of the gap to peak is workgroup prologue / epilogue / dispatch.  See DESIGN.md section 5.)
        if constexpr (BM == 256) {                  // (GEMM_SW: the one-wave-per-SIMD form of the same tile, gemm_sw.hip)
            const int sw = opt(OPT_GEMM_SW);
            // 3: where it measured ahead -- 256x256 tiles whose B operand is [K, N] (input gradients and weight gradients: +0.5 ... +2.6 % per launch;
            //    the forward layout, B = nn.Linear weight [N, K], measured 3 % behind and stays on the two-waves-per-SIMD kernel)
            const bool take = sw == 1 || (sw == 2 && g.splits == 1) || (sw == 3 && BN == 256 && !transB);
            if (take && !g.x3 && launch_sw(BM, BN, g, transA, transB, grid.x, s)) return 0;
        }
        if constexpr (BM == 224 || BM == 160) {     // 8 waves of BM x 32 (K-contiguous A only: plan_gemm)
            dim3 blk(512);
            if (transB) hipLaunchKernelGGL((gemm_dma_kernel<BM, 256, true, true, 8, 32, 2, 2, false, 8>), grid, blk, 0, s, g);
            else hipLaunchKernelGGL((gemm_dma_kernel<BM, 256, true, false, 8, 32, 2, 2, false, 8>), grid, blk, 0, s, g);
        } else if constexpr (BM == 256 && BN == 256) {
            YT_DMA(8, 32, 2, 2);                    // 8 waves of 64x128, one workgroup per CU
        } else if constexpr (BM == 256 && BN == 128) {
            YT_DMA(8, 32, 2, 2);                    // 8 waves of 64x64, one workgroup per CU (native instruction only, like 64x64)
        } else if constexpr (BM == 128 && BN == 128) {
            YT_DMA(8, 32, 2, 4);
        } else {
            YT_DMA(4, 32, 2, 2);
        }
#undef YT_DMA
        return 0;
    }
    if constexpr (BM <= 128) {
        if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true>), grid, block, 0, s, g);
        else if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false>), grid, block, 0, s, g);
        else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true>), grid, block, 0, s, g);
    }
    return 0;
}


}  // namespace ytvln

using namespace ytvln;

extern "C" int ytvln_gemm_plan(int M, int N, int K, int transA, int epilogue, int* tile_m, int* tile_n, int* splits) {
    YT_REQUIRE(tile_m && tile_n && splits && M > 0 && N > 0 && K > 0, "gemm_plan: bad argument");
    static const int bm[7] = {128, 128, 64, 256, 256, 224, 160}, bn[7] = {128, 64, 64, 128, 256, 256, 256};
    const Plan p = plan_gemm(M, N, K, epilogue, true, false, transA != 0);
    *tile_m = bm[p.tile]; *tile_n = bn[p.tile]; *splits = p.splits;
    return 0;
}

extern "C" int ytvln_gemm_plan_x3(int M, int N, int K, int transA, int epilogue, int* tile_m, int* tile_n, int* splits) {
    YT_REQUIRE(tile_m && tile_n && splits && M > 0 && N > 0 && K > 0, "gemm_plan_x3: bad argument");
    (void)transA;       // the three-term form takes the 256-row tiles with either A layout (see ytvln_gemm_f32)
    static const int bm[7] = {128, 128, 64, 256, 256, 224, 160}, bn[7] = {128, 64, 64, 128, 256, 256, 256};
    const Plan p = plan_gemm(M, N, K, epilogue, true, true);
    *tile_m = bm[p.tile]; *tile_n = bn[p.tile]; *splits = p.splits;
    return 0;
}

#if YT_GEMM_PROBE
extern "C" int ytvln_gemm_f32_probe_read(uint32_t* host128) { return hipMemcpyFromSymbol(host128, HIP_SYMBOL(g_f32_probe), 512) == hipSuccess ? 0 : -1; }
#endif
extern "C" int64_t ytvln_gemm_workspace_elems(int M, int N, int K, int epilogue) {
    const int splits = std::max(std::max(std::max(std::max(plan_splits(M, N, K, epilogue), plan_gemm(M, N, K, epilogue, true, false, false).splits),
                                                  plan_gemm(M, N, K, epilogue, true, false, true).splits),
                                         std::max(plan_gemm(M, N, K, epilogue, false, true).splits,
                                                  plan_gemm(M, N, K, epilogue, true, true).splits)),
                                1);
    // (covers either A layout and the fp32x3 plans)
    int64_t need = splits > 1 ? (int64_t)splits * M * N + (int64_t)splits * ((M + 3) / 4 * 4) : 0;      // partial tiles + partial row sums of A
    // the stream-K form of the persistent kernel (K-contiguous A; opt-in GEMM_SK): one partial tile per workgroup, sized for either tile.
    // Only while the option is on (134 MB per GEMM for a kernel that is off by default); a caller that caches the figure keys it
    // by the option (ytvln.ops._gemm does).
    if (opt(OPT_GEMM_SK) != 0 && M >= 256 && N >= 128) need = std::max<int64_t>(need, (int64_t)2 * sk_num_cus() * 256 * (N >= 256 ? 256 : 128));
    return need;
}

static int gemm_f32_impl(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                         int64_t ldc, const float* bias, float* aux, int64_t ldaux, int M, int N, int K,
                         int epilogue, float beta, float* workspace, int64_t workspace_elems, int flags, float* a_rowsum, int* rowsum_done,
                         unsigned* sk_ctl, void* stream) {
    if (rowsum_done) *rowsum_done = 0;
    YT_REQUIRE(A && B && C, "gemm: null operand");
    YT_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative size");
    YT_REQUIRE(epilogue >= YTVLN_EPI_NONE && epilogue <= YTVLN_EPI_MUL_DRELU, "gemm: bad epilogue %d", epilogue);
    YT_REQUIRE(!(epilogue >= YTVLN_EPI_MUL_DGELU) || aux, "gemm: epilogue %d needs aux", epilogue);
    YT_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "gemm: leading dimension too small");
    if (M == 0 || N == 0) return 0;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
    g.M = M; g.N = N; g.K = K; g.epilogue = epilogue; g.beta = beta;
    g.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0);
    g.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0);
    hipStream_t s = as_stream(stream);
    g.splits = 1; g.kchunk = K; g.ws = nullptr;
    g.x3 = (flags & YTVLN_GEMM_SPLIT_BF16X3) != 0;
    g.split_map = opt(OPT_GEMM_SPLIT_MAP);
    g.asum = nullptr; g.asum_ws = nullptr;
    // Fast-path legality.  With YTVLN_GEMM_A_ZERO_PADDED the caller guarantees that A's contiguous dimension is followed by
    // readable ZERO padding up to lda: a K-contiguous A may then have K % 32 != 0 (the loop runs over the rounded-up K and
    // B's k rows are clamped -- B must be [K,N]), and an M-contiguous A may have M % 4 != 0.
    const bool apad = (flags & YTVLN_GEMM_A_ZERO_PADDED) != 0;
    const int k32 = (int)cdiv(K, BK) * BK, m4 = (int)cdiv(M, 4) * 4;
    g.Kloop = K; g.ktail = 0; g.mnA = M; g.mnB = N;
    bool k_ok = (K % BK == 0);
    if (!k_ok && apad && !transA && !transB && lda >= k32) { k_ok = true; g.Kloop = k32; g.ktail = 1; }
    bool ma_ok = !transA || (M % 4 == 0 && M >= 4);
    if (!ma_ok && apad && transA && lda >= m4 && M >= 4) { ma_ok = true; g.mnA = m4; }
    g.fast = (K > 0) && k_ok && g.vecA && g.vecB && ma_ok && (!transB ? (N % 4 == 0 && N >= 4) : true) &&
             !opt(OPT_GEMM_GENERIC);
    // the 256-row tiles take either A layout: K-contiguous (forward / input gradients) and M-contiguous (weight gradients; 256x256 + split-K
    // competes with 128x128 since round 2), in the native and in the three-term form
    const bool ta_native = g.fast && transA && !g.x3;
    Plan plan = plan_gemm(M, N, K, epilogue, g.fast, g.x3 && g.fast, ta_native);
    const int want = plan.splits;
    // persistent kernel (gemm_sk.hip): needs the caller's zero-initialised control block; taken when its model cost beats the launch-per-tile plan
    if (sk_ctl && !a_rowsum) {
        SkPlan sp = plan_sk(M, N, g.Kloop, transA, epilogue, g.fast, g.x3 != 0);
        if (sp.use && !sp.dp && (!workspace || workspace_elems < sk_workspace_elems(sp))) sp.use = 0;
        if (sp.use && (opt(OPT_GEMM_SK) >= 2 || sp.cost < 0.98 * plan_cost(M, N, K, plan.tile, plan.splits, epilogue, false, ta_native))) {
            sk_launch(g, sp, transB, workspace, sk_ctl, s);
            YT_LAUNCH_CHECK("gemm_f32 (persistent)");
            return 0;
        }
    }
    // row sums of op(A) ride on launches that take the LDS-DMA main loop with an M-contiguous fp32 A and no K tail (a clamped tail row would
    // be counted twice); everything else reports "not done" and the caller runs ytvln_colsum_f32
    const bool asum_ok = a_rowsum && g.fast && transA && !g.x3 && !g.ktail && K % BK == 0;
    const int64_t m4r = (M + 3) / 4 * 4;
    if (want > 1 && workspace && workspace_elems >= (int64_t)want * M * N + (asum_ok ? (int64_t)want * m4r : 0)) {
        g.kchunk = (int)cdiv(cdiv(K, want), BK) * BK;
        g.splits = (int)cdiv(K, g.kchunk);
        g.ws = workspace;
        if (asum_ok) { g.asum = a_rowsum; g.asum_ws = workspace + (int64_t)g.splits * M * N; }
    } else if (want > 1) {                 // no workspace supplied: best unsplit plan
        plan.splits = 1;
        double bt = 1e30;
        for (int tile = 0; tile < 3; ++tile) {
            const double t = plan_cost(M, N, K, tile, 1, 0, g.x3 != 0, ta_native);
            if (t < bt * 0.98) { bt = t; plan.tile = tile; }
        }
    }
    if (g.splits == 1) g.kchunk = std::max(g.kchunk, g.Kloop);
    if (g.splits == 1 && asum_ok) g.asum = a_rowsum;
    if (rowsum_done) *rowsum_done = g.asum != nullptr;
    if (g.splits > 1) {
        if (plan.tile == 4) launch_tile<256, 256>(g, transA, transB, s);
        else if (plan.tile == 5 && g.fast && !transA && !g.x3) launch_tile<224, 256>(g, transA, transB, s);
        else if (plan.tile == 6 && g.fast && !transA && !g.x3) launch_tile<160, 256>(g, transA, transB, s);
        else launch_tile<128, 128>(g, transA, transB, s);
        const int64_t total = (int64_t)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 1024), 2048)), dim3(256), 0, s, workspace, C,
                           ldc, bias, M, N, g.splits, beta, (const float*)g.asum_ws, g.asum);
    } else {
        g.splits = 1;
        if (plan.tile == 4) launch_tile<256, 256>(g, transA, transB, s);
        else if (plan.tile == 5 && g.fast && !transA && !g.x3) launch_tile<224, 256>(g, transA, transB, s);
        else if (plan.tile == 6 && g.fast && !transA && !g.x3) launch_tile<160, 256>(g, transA, transB, s);
        else if (plan.tile >= 5) launch_tile<256, 256>(g, transA, transB, s);
        else if (plan.tile == 3) launch_tile<256, 128>(g, transA, transB, s);
        else if (plan.tile == 0) launch_tile<128, 128>(g, transA, transB, s);
        else if (plan.tile == 1) launch_tile<128, 64>(g, transA, transB, s);
        else launch_tile<64, 64>(g, transA, transB, s);
    }
    YT_LAUNCH_CHECK("gemm_f32");
    return 0;
}

extern "C" int ytvln_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                              int64_t ldc, const float* bias, float* aux, int64_t ldaux, int M, int N, int K,
                              int epilogue, float beta, float* workspace, int64_t workspace_elems, int flags, void* stream) {
    return gemm_f32_impl(A, lda, transA, B, ldb, transB, C, ldc, bias, aux, ldaux, M, N, K, epilogue, beta, workspace, workspace_elems, flags,
                         nullptr, nullptr, nullptr, stream);
}

extern "C" int ytvln_gemm_f32_rowsum(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                                     int64_t ldc, const float* bias, float* aux, int64_t ldaux, int M, int N, int K,
                                     int epilogue, float beta, float* workspace, int64_t workspace_elems, int flags, float* a_rowsum,
                                     int* rowsum_done, void* stream) {
    YT_REQUIRE(a_rowsum && rowsum_done, "gemm_f32_rowsum: null row-sum output");
    return gemm_f32_impl(A, lda, transA, B, ldb, transB, C, ldc, bias, aux, ldaux, M, N, K, epilogue, beta, workspace, workspace_elems, flags,
                         a_rowsum, rowsum_done, nullptr, stream);
}

extern "C" int ytvln_gemm_f32_sk(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                                 int64_t ldc, const float* bias, float* aux, int64_t ldaux, int M, int N, int K,
                                 int epilogue, float beta, float* workspace, int64_t workspace_elems, int flags, float* a_rowsum,
                                 int* rowsum_done, uint32_t* sk_ctl, void* stream) {
    YT_REQUIRE((a_rowsum == nullptr) == (rowsum_done == nullptr), "gemm_f32_sk: a_rowsum and rowsum_done go together");
    return gemm_f32_impl(A, lda, transA, B, ldb, transB, C, ldc, bias, aux, ldaux, M, N, K, epilogue, beta, workspace, workspace_elems, flags,
                         a_rowsum, rowsum_done, sk_ctl, stream);
}

extern "C" int ytvln_gemm_sk_plan(int M, int N, int K, int transA, int epilogue, int* use, int* tile_m, int* tile_n, int* whole_tiles,
                                  int* workgroups) {
    YT_REQUIRE(use && tile_m && tile_n && whole_tiles && workgroups && M > 0 && N > 0 && K > 0, "gemm_sk_plan: bad argument");
    const SkPlan sp = plan_sk(M, N, (int)cdiv(K, BK) * BK, transA, epilogue, true, false);
    const Plan plan = plan_gemm(M, N, K, epilogue, true, false, transA != 0);
    *use = sp.use && (opt(OPT_GEMM_SK) >= 2 || sp.cost < 0.98 * plan_cost(M, N, K, plan.tile, plan.splits, epilogue, false, transA != 0));
    *tile_m = sp.tile == 0 ? 128 : 256; *tile_n = sp.tile == 4 ? 256 : 128; *whole_tiles = sp.dp; *workgroups = sp.G;
    return 0;
}

#endif  // YT_GEMM_X3_TU
