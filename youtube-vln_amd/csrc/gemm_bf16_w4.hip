// bf16 GEMM, FOUR waves of 128x128 on a 256x256 tile: one wave per SIMD with the whole 512-entry register file (run-time option
// GEMM_BF16_FORM = 4).  Same arithmetic, operand layouts, LDS images, epilogues and split-K as gemm_bf16_kernel (gemm_bf16.hip).
//
// Why: the round-6 phase timeline of the eight-wave kernel (tools/gemm_bf16_probe.py, profiles/round6_gemm_bf16_probe.log) shows its load
// phases -- 12 ds_read_b128 per wave while the other group's LDS-DMA pieces land -- taking 670-730 cycles against 580-616 for the 16 matrix
// instructions they are meant to hide under: the kernel is bound by LDS bandwidth (192 KB of fragment reads + 64 KB of DMA writes per 64-deep
// k-tile), not by the matrix pipe, not by DMA latency (forms 1-3 moved the DMA issue and deepened the rings: 0-10 % slower).  A wave's fragment
// traffic per flop is set by its tile: (64 + 128) columns per 64 x 128 outputs there, (128 + 128) per 128 x 128 here = 2/3 of the LDS reads for
// the same matrix work (128 KB per k-tile).  The price is one wave per SIMD: nobody else covers this wave's latencies, so its own stream is
// software-pipelined -- per 64-deep k-tile four steps of 16 matrix instructions (512 cycles of the pipe each), and under each step's matrix
// instructions the 8 fragment reads of the NEXT step and a share of the LDS-DMA of later tiles:
//     step 0   reads(step 1)  + A(kt+2) pieces 0-3      16 MFMA
//     step 1   reads(step 2)  + A(kt+2) pieces 4-7      16 MFMA
//     step 2   reads(step 3)                            16 MFMA       then: this wave's reads of tile kt have returned (lgkmcnt(0)), its pieces of
//                                                                     tile kt+1 have landed (counted vmcnt)  -> the k-tile's ONE barrier
//     step 3   B(kt+2) pieces 0-7 + reads(step 0 of tile kt+1)        16 MFMA
// Rings as in gemm_bf16_kernel: A three slots (tile kt+2 requested during tile kt), B two (tile kt+2 requested behind the barrier of tile kt,
// into the slot tile kt has just left).  Hazards: behind the barrier of tile kt every wave has finished reading tile kt (slots A[kt % 3] and
// B[kt % 2] are free) and tile kt+1 is complete in LDS.  A(kt+2) -> slot (kt+2) % 3 = (kt-1) % 3, free since the barrier of tile kt-1: issued
// in steps 0-1 of tile kt.  B(kt+2) -> slot kt % 2: issued in step 3 of tile kt.  At the wait in front of the barrier of tile kt the wave's
// queue holds, oldest first, ... B(kt+1) [step 3 of kt-1], A(kt+2) [steps 0-1 of kt]: vmcnt(NI_A) = everything up to B(kt+1) landed; A(kt+1)
// is older still.
#define YT_BF16_SHARED_ONLY 1
#include "gemm_bf16.hip"

#ifndef YT_W4_MEASURE
#define YT_W4_MEASURE 0
#endif

namespace ytvln {

template <int N>
__device__ __forceinline__ void w4_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// ABL (measurement builds, wrong results by construction): 1 = no LDS-DMA inside the loop, 2 = no fragment reads inside the loop, 3 = neither
template <bool A_KC, bool B_KC, typename CT, int ABL = 0, bool PROBE = false>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(const BfArgs g) {
    constexpr int BM = 256, BN = 256, NW = 4;
    using TA = BfTile<BM, A_KC, NW>;
    using TB = BfTile<BN, B_KC, NW>;
    constexpr int TM = 4, TN = 4;                                           // 128 x 128 per wave
    constexpr int SA = BM * KT * 2, SB = BN * KT * 2;
    constexpr int NSA = 3, NSB = 2;
    constexpr int NIA = TA::NI, NIB = TB::NI;                               // 8 + 8 LDS-DMA pieces per wave and k-tile
    static_assert(NIA == 8 && NIB == 8, "piece schedule below assumes 8 + 8");
    __shared__ __attribute__((aligned(16))) char smem[NSA * SA + NSB * SB];
    char* const ringA = smem;
    char* const ringB = smem + NSA * SA;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * 128, wn0 = (wave & 1) * 128;
    const BfCoord tc = bf_decode(blockIdx.x, g.tiles_m, g.tiles_n, g.splits);
    const int m0 = tc.m * BM, n0 = tc.n * BN;
    const int kbeg = tc.split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + KT - 1) / KT;
    const int nfull = (kend - kbeg) / KT;

    const bf16_t* pa[NIA];
    const bf16_t* pb[NIB];
#pragma unroll
    for (int i = 0; i < NIA; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg, wave * NIA + i, lane);
#pragma unroll
    for (int i = 0; i < NIB; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg, wave * NIB + i, lane);
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

    // LDS-DMA piece i of A / B tile `kt` into the ring slot the tile owns (whole k-tiles only: the launcher keeps K % 64 != 0 off this kernel,
    // so the hot loop carries no register-staged tail branch)
    // The loop body is BRANCH-FREE: past the last tile the same requests are issued once more with a zero pointer step (they re-fetch the last
    // tile into a ring slot nobody reads any more) -- with an `if (tile exists)` around every piece each step became its own chain of basic
    // blocks and the compiler's wait insertion fell back to lgkmcnt(0) at every join, i.e. in front of every step's first matrix instruction.
    auto dmaA = [&](int kt, auto ic, int64_t step) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (ABL & 1) { if (kt >= 2) return; }
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[i], (lds_ptr_t)(ringA + (kt % NSA) * SA + (wave * NIA + i) * 1024), 16, 0, 0);
        pa[i] += step;
    };
    auto dmaB = [&](int kt, auto ic, int64_t step) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (ABL & 1) { if (kt >= 2) return; }
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[i], (lds_ptr_t)(ringB + (kt % NSB) * SB + (wave * NIB + i) * 1024), 16, 0, 0);
        pb[i] += step;
    };

    bf16x8 fra[2][TM], frb[2][TN];          // fragments of the step being multiplied / of the next step
    auto read_step = [&](auto bufc, const char* __restrict__ As, const char* __restrict__ Bs, int s) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) fra[buf][i] = fa.get(As, i, half, s);
#pragma unroll
        for (int j = 0; j < TN; ++j) frb[buf][j] = fb.get(Bs, j, half, s);
    };
    // One step: the 16 matrix instructions on fragment buffer `buf`; WHICH = 0 nothing else, 1 / 2: A pieces 0-3 / 4-7 of tile kt_dma behind
    // matrix instructions 1, 5, 9, 13; 3: B pieces 0-7 behind every second one.  The order is pinned (sched_barrier): the compiler otherwise
    // sinks the next step's fragment reads down to their first use and the wave waits out the LDS latency with an idle matrix pipe.
    // One step: the 16 matrix instructions on fragment buffer `buf`, with ONE other memory instruction per gap (the ablation builds showed that
    // whatever this lone wave issues in a burst is paid in full -- 8 fragment reads back to back cost it 190 cycles of idle matrix pipe, the
    // pipe executes 32 cycles per instruction): behind matrix instructions 0-7 the 8 fragment reads of the NEXT step (4 A, 4 B) into the other
    // buffer, behind 8-15 this step's LDS-DMA pieces -- WHICH = 0 none, 1 / 2: A pieces 0-3 / 4-7 of tile kt_dma (every second gap), 3: B pieces
    // 0-7.  The order is pinned (sched_barrier): left alone the compiler sinks the reads down to their first use.
    auto mfma_step = [&](auto bufc, auto whichc, int kt_dma, int64_t stepA, int64_t stepB, const char* __restrict__ rAs, const char* __restrict__ rBs,
                         int rs) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value, WHICH = decltype(whichc)::value, nb = 1 - buf;
        __builtin_amdgcn_sched_barrier(0);
        static_for<TM * TN>([&](auto ic) {
            constexpr int idx = decltype(ic)::value, i = idx / TN, j = idx % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[buf][i], frb[buf][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (idx < 8) {
                if constexpr (ABL & 2) {
                    if constexpr (idx < 4) asm volatile("" : "+v"(fra[nb][idx]));
                    else asm volatile("" : "+v"(frb[nb][idx - 4]));
                } else {
                    if constexpr (idx < 4) fra[nb][idx] = fa.get(rAs, idx, half, rs);
                    else frb[nb][idx - 4] = fb.get(rBs, idx - 4, half, rs);
                }
            } else if constexpr (WHICH == 1 || WHICH == 2) {
                if constexpr (idx % 2 == 0) dmaA(kt_dma, std::integral_constant<int, (WHICH - 1) * 4 + (idx - 8) / 2>{}, stepA);
            } else if constexpr (WHICH == 3) {
                dmaB(kt_dma, std::integral_constant<int, idx - 8>{}, stepB);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using W3 = std::integral_constant<int, 3>;

    // prologue: A(0), B(0), A(1), B(1) in flight (a single-tile contraction requests its only tile twice); wait for tile 0; first fragments
    {
        const int64_t s0a = nk > 1 ? sa : 0, s0b = nk > 1 ? sb : 0, s1a = nk > 2 ? sa : 0, s1b = nk > 2 ? sb : 0;
        static_for<NIA>([&](auto ic) { dmaA(0, ic, s0a); });
        static_for<NIB>([&](auto ic) { dmaB(0, ic, s0b); });
        static_for<NIA>([&](auto ic) { dmaA(1, ic, s1a); });
        static_for<NIB>([&](auto ic) { dmaB(1, ic, s1b); });
    }
    w4_wait<NIA + NIB>();
    __builtin_amdgcn_s_barrier();
    read_step(B0{}, ringA, ringB, 0);

    // PROBE (ytvln_gemm_bf16_probe, form 44): every wave of workgroup probe_block stamps s_memtime for k-tiles 8..23 at: 0 steps 0-2 issued (in
    // front of the wait), 1 wait passed, 2 barrier released, 3 step 3 issued -> probe[64 * wave + 4 * (kt - 8) + point]
    uint32_t ts = 0;
    const bool probing = PROBE && g.probe != nullptr && (int)blockIdx.x == g.probe_block;
    auto stamp = [&](int kt, int p) __attribute__((always_inline)) {
        if constexpr (PROBE) {
            if (probing && kt >= 8 && kt < 24) {
                const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime();
                ts = lane == (kt - 8) * 4 + p ? now : ts;
            }
        }
    };
    for (int kt = 0; kt < nk; ++kt) {
        const char* As = ringA + (kt % NSA) * SA;
        const char* Bs = ringB + (kt % NSB) * SB;
        const int64_t stA = kt + 3 < nk ? sa : 0, stB = kt + 3 < nk ? sb : 0;          // the requests below are for tile kt+2: step on only if kt+3 exists
        // ---- step 0 (the next step's reads sit BEHIND the first matrix instructions: see mfma_step)
        mfma_step(B0{}, W1{}, kt + 2, stA, stB, As, Bs, 1);
        // ---- step 1
        mfma_step(B1{}, W2{}, kt + 2, stA, stB, As, Bs, 2);
        // ---- step 2
        mfma_step(B0{}, W0{}, 0, 0, 0, As, Bs, 3);
        // the k-tile's barrier: this wave's reads of tile kt have returned; its pieces of tile kt+1 (A: requested one tile ago, B: behind the
        // previous barrier) have landed -- A(kt+2), requested in steps 0-1, stays in flight
        stamp(kt, 0);
        w4_wait<NIA>();
        __builtin_amdgcn_sched_barrier(0);
        stamp(kt, 1);
        __builtin_amdgcn_s_barrier();
        stamp(kt, 2);
        // ---- step 3
        mfma_step(B1{}, W3{}, kt + 2, stA, stB, ringA + ((kt + 1) % NSA) * SA, ringB + ((kt + 1) % NSB) * SB, 0);
        stamp(kt, 3);
    }
    if constexpr (PROBE) {
        if (probing) g.probe[wave * 64 + lane] = ts;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing requests must have landed before this workgroup's LDS is handed on
    // The epilogue, one 32-row block of the wave's tile at a time: the accumulators live in the accumulation half of the register file and
    // every epilogue value has to pass through the other half.  Left to itself hipcc moves all 256 across in front of the epilogue switch and
    // spills 60-100 of them (26 us of fixed cost per round, measured); here each block's 64 values are read out explicitly just before use.
    // wide stores of a bf16 C through this wave's slice of the idle LDS where the epilogue allows (gemm_bf16.hip: bf_wide_*)
    constexpr bool C16 = std::is_same<CT, bf16_t>::value;
    BfArgs gw = g; gw.wide_stores = 1;          // (this form always takes the wide path where it is legal: its lone wave per SIMD is store-issue bound)
    const bool wide = C16 && bf_wide_ok<TM, TN>(gw, m0 + wm0, n0 + wn0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every wave is done with the operand rings before any wave reuses LDS (`wide` differs between waves
                                                             // of an edge tile, so the barrier is unconditional)
    // block i's 64 accumulators, read out of the accumulation registers just before use (left to itself hipcc moves all 256 across in front of the
    // epilogue switch and spills 60-100 of them: 26 us of fixed cost per round, measured)
    using Blk = BfBlk<TN>;
    auto take = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        Blk b;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i][j][r]));
                b.v[j][r] = x;
            }
        return b;
    };
    if constexpr (C16) {
        if (wide) {
            char* const tb = smem + wave * (TM * bf_wide_bytes<TN>());
            bf_wide_tile<TM, TN>(g, [&](auto ic) { return take(ic); }, tb, m0 + wm0, n0 + wn0, lane);
            return;
        }
    }
    static_for<TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        Blk b = take(ic);
        f32x16 blk[1][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) blk[0][j] = b.v[j];
        bf_epilogue<1, TN, CT>(g, blk, m0 + wm0 + 32 * i, n0 + wn0, l31, half, tc.split);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Instantiated for the launches this form is for: a contraction-contiguous A (forward and input-gradient products) with a bf16 C.  The weight
// gradients (k-major A, fp32 C, split-K) stay on gemm_bf16_kernel, where they already run at 1.4-2.9x the vendor library.
bool bf_launch_w4(const BfArgs& g, int transA, int transB, hipStream_t s) {
    if (transA || g.K % KT != 0 || g.splits != 1) return false;
    const dim3 grid((unsigned)(g.ntiles * g.splits)), blk(256);
#if YT_W4_MEASURE
    // 41 / 42 / 43: the ablation builds, 44: the stamped one (forward layout only).  Compiled only with -DYT_W4_MEASURE=1 (they quadruple this
    // file's 4-minute compile and 41-43 give wrong results by construction, so the shipped library does not carry them; there 41-44 run as form 4).
    const int abl = opt(OPT_GEMM_BF16_FORM) - 40;
    if (transB && abl == 1) { hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, true, bf16_t, 1>), grid, blk, 0, s, g); return true; }
    if (transB && abl == 2) { hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, true, bf16_t, 2>), grid, blk, 0, s, g); return true; }
    if (transB && abl == 4) { hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, true, bf16_t, 0, true>), grid, blk, 0, s, g); return true; }
    if (transB && abl == 3) { hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, true, bf16_t, 3>), grid, blk, 0, s, g); return true; }
#endif
    if (transB) hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, true, bf16_t>), grid, blk, 0, s, g);
    else hipLaunchKernelGGL((gemm_bf16_w4_kernel<true, false, bf16_t>), grid, blk, 0, s, g);
    return true;
}

}  // namespace ytvln
