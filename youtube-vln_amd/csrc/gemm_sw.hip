// fp32 GEMM, ONE wave per SIMD: the software-pipelined form of the LDS-DMA kernel (round 3's gemm_sw_kernel, taken out in round 4 when the boxes of
// that round clocked down under it, back in round 5 as its own translation unit: this round's boxes hold 2.39 GHz under a full GEMM).  Four waves
// per workgroup (2 x 2), one workgroup per CU, up to 512 registers per lane (accumulators in the AGPR half); a wave's operand fragments are
// double-buffered in registers at k-group granularity and every `ds_read_b128` / LDS-DMA issue is pinned behind a matrix instruction of the SAME
// wave (`sched_barrier` after each step: the order is the schedule) -- while one wave of a SIMD streams matrix instructions the OTHER wave of that
// SIMD only gets about one instruction issued per matrix instruction, which is what holds the two-waves-per-SIMD kernel at 0.93 of the pipe's cycles
// (this form: 0.950-0.955; LABNOTES 5b).  Epilogue through LDS (the ring is free by then): 16-byte row stores, bias / activation / beta in a
// run-time row loop; same arithmetic per element as the direct epilogue (bit-identical results).  Run-time option GEMM_SW.
#include "gemm_tiles.h"

namespace ytvln {

// Epilogue of the big-wave-tile kernels (gemm_sw_kernel): the wave's accumulators go through LDS (the operand ring is free by then) so that
// the output leaves as 16-byte row segments -- one global_store_dwordx4 per lane covers two (TN = 4) or four (TN = 2) 512 / 256-byte row
// pieces per wave instruction instead of sixteen 4-byte stores per 32 x 32 sub-tile -- and so that bias / activation / beta work runs in a
// small run-time loop over rows instead of a fully unrolled, ten-times specialised store sequence (the unrolled form of a 4 x 4 sub-tile
// wave is ~1 MB of code per kernel and spills).  `wlds`: this wave's private LDS region, 64 x (32 TN) floats; two passes for TM = 4.
// Same arithmetic per element as epilogue_body (bias add, then activation, then beta * old), so results are bit-identical.
// (`stage(PASS)` writes rows [PROWS * pass, PROWS * (pass + 1)) of the wave tile into wlds[row][W] -- the only part that knows the accumulator layout)
template <int W, int PROWS, int NPASS, class Stage>
__device__ __forceinline__ void epilogue_lds_rows(const GemmArgs& g, float* __restrict__ wlds, int row0, int col0, int lane, int split, Stage&& stage) {
    constexpr int W4 = W / 4;                           // 16-byte groups per staged row
    constexpr int RPI = 64 / W4;                        // rows covered by one wave-wide 16-byte access
    const bool partial = g.splits > 1;
    float* const Cb = partial ? g.ws + (int64_t)split * g.M * g.N : g.C;
    const int64_t ldc = partial ? g.N : g.ldc;
    const int epi = partial ? YTVLN_EPI_NONE : g.epilogue;
    const float beta = partial ? 0.f : g.beta;
    const float* bias = partial ? nullptr : g.bias;
    const int c4 = lane % W4, rsub = lane / W4;
    const int col = col0 + 4 * c4;
    const bool vec = (g.N & 3) == 0 && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(Cb) & 15) == 0 &&
                     (epi == YTVLN_EPI_NONE || epi == YTVLN_EPI_RELU || g.aux == nullptr ||
                      ((g.ldaux & 3) == 0 && (reinterpret_cast<uintptr_t>(g.aux) & 15) == 0));
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[u] = col + u < g.N ? bias[col + u] : 0.f;
    }
    static_for<NPASS>([&](auto PASS) __attribute__((always_inline)) {
        stage(PASS);                                         // (compile-time pass index: a run-time one would move the accumulators to scratch memory)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int rbase = row0 + PROWS * decltype(PASS)::value;
        auto rows = [&](auto EPI_TAG) __attribute__((always_inline)) {
            constexpr int EPI = decltype(EPI_TAG)::value;
            constexpr bool AUXLD = EPI == YTVLN_EPI_MUL_DGELU || EPI == YTVLN_EPI_MUL_DRELU;
            constexpr int NIT = PROWS / RPI, CH = 8;             // CH row accesses at a time: their global loads are in flight together
            static_assert(NIT % CH == 0, "chunking");
            for (int it0 = 0; it0 < NIT; it0 += CH) {
                float4 t[CH], o[CH], a[CH];
                if (vec) {
                    if (beta != 0.f) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const int row = rbase + (it0 + c) * RPI + rsub;
                            o[c] = (row < g.M && col < g.N) ? *reinterpret_cast<const float4*>(Cb + (int64_t)row * ldc + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    if constexpr (AUXLD) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const int row = rbase + (it0 + c) * RPI + rsub;
                            a[c] = (row < g.M && col < g.N) ? *reinterpret_cast<const float4*>(g.aux + (int64_t)row * g.ldaux + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) t[c] = *reinterpret_cast<const float4*>(wlds + ((it0 + c) * RPI + rsub) * W + 4 * c4);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int row = rbase + (it0 + c) * RPI + rsub;
                    if (row >= g.M || col >= g.N) continue;
                    float v[4] = {t[c].x + bv[0], t[c].y + bv[1], t[c].z + bv[2], t[c].w + bv[3]};
                    float* cp = Cb + (int64_t)row * ldc + col;
                    float* xp = (EPI == YTVLN_EPI_GELU || AUXLD) ? g.aux + (int64_t)row * g.ldaux + col : nullptr;
                    if (vec) {
                        if constexpr (EPI == YTVLN_EPI_GELU) {
                            if (g.aux) *reinterpret_cast<float4*>(xp) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = gelu_erf(v[u]);
                        } else if constexpr (EPI == YTVLN_EPI_RELU) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
                        } else if constexpr (EPI == YTVLN_EPI_MUL_DGELU) {
                            v[0] *= dgelu_erf(a[c].x); v[1] *= dgelu_erf(a[c].y); v[2] *= dgelu_erf(a[c].z); v[3] *= dgelu_erf(a[c].w);
                        } else if constexpr (EPI == YTVLN_EPI_MUL_DRELU) {
                            v[0] = a[c].x > 0.f ? v[0] : 0.f; v[1] = a[c].y > 0.f ? v[1] : 0.f; v[2] = a[c].z > 0.f ? v[2] : 0.f; v[3] = a[c].w > 0.f ? v[3] : 0.f;
                        }
                        if (beta != 0.f) { v[0] += beta * o[c].x; v[1] += beta * o[c].y; v[2] += beta * o[c].z; v[3] += beta * o[c].w; }
                        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (col + u >= g.N) continue;
                            float x = v[u];
                            if constexpr (EPI == YTVLN_EPI_GELU) { if (g.aux) xp[u] = x; x = gelu_erf(x); }
                            else if constexpr (EPI == YTVLN_EPI_RELU) x = fmaxf(x, 0.f);
                            else if constexpr (EPI == YTVLN_EPI_MUL_DGELU) x *= dgelu_erf(xp[u]);
                            else if constexpr (EPI == YTVLN_EPI_MUL_DRELU) x = xp[u] > 0.f ? x : 0.f;
                            if (beta != 0.f) x += beta * cp[u];
                            cp[u] = x;
                        }
                    }
                }
            }
        };
        switch (epi) {
            case YTVLN_EPI_GELU: rows(std::integral_constant<int, YTVLN_EPI_GELU>{}); break;
            case YTVLN_EPI_RELU: rows(std::integral_constant<int, YTVLN_EPI_RELU>{}); break;
            case YTVLN_EPI_MUL_DGELU: rows(std::integral_constant<int, YTVLN_EPI_MUL_DGELU>{}); break;
            case YTVLN_EPI_MUL_DRELU: rows(std::integral_constant<int, YTVLN_EPI_MUL_DRELU>{}); break;
            default: rows(std::integral_constant<int, YTVLN_EPI_NONE>{}); break;
        }
    });
}

template <int TM, int TN>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& g, f32x16 (&acc)[TM][TN], float* __restrict__ wlds, int row0, int col0, int lane, int split) {
    constexpr int W = 32 * TN, IB = TM >= 2 ? 2 : 1;    // floats per staged row; 32-row blocks per pass
    const int l31 = lane & 31, half = lane >> 5;
    epilogue_lds_rows<W, 32 * IB, TM / IB>(g, wlds, row0, col0, lane, split, [&](auto PASS) __attribute__((always_inline)) {
        constexpr int i0 = decltype(PASS)::value * IB;
#pragma unroll
        for (int ii = 0; ii < IB; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wlds[(32 * ii + (r & 3) + 8 * (r >> 2) + 4 * half) * W + 32 * j + l31] = acc[i0 + ii][j][r];
    });
}

// ---- one wave per SIMD, software-pipelined main loop (round 3) -----------------------------------------------------------------------
// What the round-3 probes measured (tools/lab/gen_mfma_lds_bench.py, profiles/round3_pp_probes.log): while one wave of a SIMD
// streams fp32 matrix instructions, the OTHER wave of that SIMD gets about one vector-memory / LDS / VALU instruction issued per matrix
// instruction (64 cycles) -- a 35-instruction operand-load phase takes ~2100-2300 cycles beside a 2048-cycle matrix phase, which is what
// holds gemm_dma_kernel (and the ping-pong form above) at 0.80-0.90 of the matrix peak.  The same instructions placed INSIDE the
// matrix-issuing wave's own stream cost ~3 cycles per ds_read_b128 and ~11 per LDS-DMA piece.  Hence this form: four waves per workgroup,
// one per SIMD, one workgroup per CU, up to 512 registers per lane; a wave's operand fragments are double-buffered in registers at
// k-group granularity (4 k per group, 4 groups per 32-deep k-tile) and every fragment read / DMA issue is interleaved one behind a matrix
// instruction (sched_group_barrier).  ONE workgroup barrier per k-tile, between k-groups 2 and 3:
//   group 0..2 of tile t : matrix instructions of group g on F[g & 1]  ||  reads of group g + 1 -> F[(g + 1) & 1]
//   before the barrier   : own DMA pieces of tile t + 1 retired (counted vmcnt), own reads of (t, group 3) retired (lgkmcnt(0))
//   group 3 of tile t    : matrix instructions on F[1]  ||  DMA of tile t + NS into the slot of tile t (every wave is done reading it)
//                                                       ||  reads of (t + 1, group 0) -> F[0]
// so NS slots keep NS - 1 tiles of DMA in flight under a full tile of matrix work each.
// Cross-XCD exchange of stream-K partial tiles without cache-wide fences (guide, section 5.7): write-through (sc1) 16-byte stores,
// vmcnt drain, workgroup barrier, ONE relaxed agent-scope flag store; the owner polls the flag relaxed (an acquire poll would invalidate
// its L1 on every iteration) and reads the slab with sc1 loads.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(float* p, v4f v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ v4f load_sc1(const float* p) {      // the caller waits (s_waitcnt vmcnt(0)) before using the value
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// The main loop of the one-wave-per-SIMD kernels as a callable: k-tiles [kbeg / 32, kbeg / 32 + nk) of the block tile at (m0, n0) are
// accumulated into `acc` (zeroed by the caller).  On return no DMA is in flight and this wave has read everything it needed from the ring;
// the caller puts a workgroup barrier between two calls (and before it reuses the ring as epilogue staging space).
template <int BM, int BN, bool A_KC, bool B_KC, int NS>
struct SwLoop {
    static constexpr int NW = 4, KB = 32;
    using TA = DmaTile<BM, A_KC, NW, KB>;
    using TB = DmaTile<BN, B_KC, NW, KB>;
    static constexpr int TM = BM / 2 / 32, TN = BN / 2 / 32;  // waves 2 (m) x 2 (n)
    static constexpr int SA = BM * KB, SB = BN * KB, STAGE = SA + SB;
    static constexpr int NPT = TA::NI + TB::NI;
    static constexpr int NMF = TM * TN * 4;                    // matrix instructions per k-group
    static_assert(NS >= 2 && NS <= 4 && NS * STAGE * 4 <= 160 * 1024, "ring does not fit the LDS");

    __device__ static __forceinline__ void run(const GemmArgs& g, float* __restrict__ smem, int m0, int n0, int kbeg, int nk, bool tail_here,
                                               f32x16 (&acc)[TM][TN], float (&asum)[TM], bool do_asum, int wave, int lane,
                                               int unused_ = 0) {
        const int l31 = lane & 31, half = lane >> 5;
        const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
        const float* pa[TA::NI];
        const float* pb[TB::NI];
    #pragma unroll
        for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg, wave, lane, i, 0x7fffffff);
    #pragma unroll
        for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg, wave, lane, i, 0x7fffffff);
        const int64_t sa = TA::step(g.lda), sb = TB::step(g.ldb);

        int st_in = 0;
        auto issue = [&](int kt) {
            float* As = smem + st_in * STAGE;
            float* Bs = As + SA;
            st_in = (st_in + 1 == NS) ? 0 : st_in + 1;
            if (tail_here && kbeg + (kt + 1) * KB > g.K) {
    #pragma unroll
                for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg + kt * KB, wave, lane, i, g.K - 1);
    #pragma unroll
                for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg + kt * KB, wave, lane, i, g.K - 1);
            }
    #pragma unroll
            for (int i = 0; i < TA::NI; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[i], (lds_ptr_t)(As + (wave * TA::NI + i) * 256), 16, 0, 0);
                pa[i] += sa;
            }
    #pragma unroll
            for (int i = 0; i < TB::NI; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[i], (lds_ptr_t)(Bs + (wave * TB::NI + i) * 256), 16, 0, 0);
                pb[i] += sb;
            }
        };
        auto retire = [&](int left) {         // wait for this wave's oldest outstanding tile; `left` younger tiles stay in flight
            if (NS >= 4 && left >= 2) wait_vmcnt<2 * NPT>();
            else if (NS >= 3 && left >= 1) wait_vmcnt<NPT>();
            else wait_vmcnt<0>();
        };

        float4 fa[2][TM], fb[2][TN];
        constexpr int NFR = TM + TN;                        // fragments per k-group
        static_assert(NFR + NPT <= NMF, "more loads than matrix instructions in a k-group");
        // (every register-array index below is a compile-time constant: a run-time index makes hipcc move the fragment arrays to scratch LDS)
        // fragment R of k-group sg: the A fragments first, then the B fragments (one ds_read_b128, or four ds_read_b32 of a k-major image)
        auto rd1 = [&](auto BUF, auto R, const float* __restrict__ As, const float* __restrict__ Bs, int sg) __attribute__((always_inline)) {
            constexpr int buf = decltype(BUF)::value, r = decltype(R)::value;
            if constexpr (r < TM) fa[buf][r] = TA::frag(As, wm0, r, l31, half, sg);
            else fb[buf][r - TM] = TB::frag(Bs, wn0, r - TM, l31, half, sg);
        };
        auto mma1 = [&](auto BUF, auto K) __attribute__((always_inline)) {      // matrix instruction K of a group: k-major, accumulators alternate
            constexpr int buf = decltype(BUF)::value, k = decltype(K)::value;
            constexpr int c = k / (TM * TN), i = (k / TN) % TM, j = k % TN;
            const float av = c == 0 ? fa[buf][i].x : c == 1 ? fa[buf][i].y : c == 2 ? fa[buf][i].z : fa[buf][i].w;
            const float bv = c == 0 ? fb[buf][j].x : c == 1 ? fb[buf][j].y : c == 2 ? fb[buf][j].z : fb[buf][j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
        };
        // One k-group: matrix instruction k on F[CUR], then (pinned behind it) fragment k of the NEXT group into F[CUR ^ 1], then -- group 3 of a
        // steady tile -- DMA piece k - NFR of tile kt + NS.  sched_barrier(0) after every step: the order IS the schedule.
        auto group = [&](auto CUR, auto DMA_TAG, const float* __restrict__ As, const float* __restrict__ Bs, int sg_next, bool reads, float* Ad) __attribute__((always_inline)) {
            constexpr int cur = decltype(CUR)::value;
            constexpr bool DMA = decltype(DMA_TAG)::value;
            if constexpr (!A_KC) {
                if (do_asum) {
    #pragma unroll
                    for (int i = 0; i < TM; ++i) asum[i] += (fa[cur][i].x + fa[cur][i].y) + (fa[cur][i].z + fa[cur][i].w);
                }
            }
            static_for<NMF>([&](auto K) __attribute__((always_inline)) {
                constexpr int k = decltype(K)::value;
                mma1(CUR, K);
                if constexpr (k < NFR) {
                    if (reads) rd1(std::integral_constant<int, cur ^ 1>{}, K, As, Bs, sg_next);
                }
                if constexpr (DMA && k >= NFR && k - NFR < TA::NI) {
                    constexpr int d = k - NFR;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[d], (lds_ptr_t)(Ad + (wave * TA::NI + d) * 256), 16, 0, 0);
                    pa[d] += sa;
                } else if constexpr (DMA && k >= NFR + TA::NI && k - NFR < NPT) {
                    constexpr int e = k - NFR - TA::NI;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[e], (lds_ptr_t)(Ad + SA + (wave * TB::NI + e) * 256), 16, 0, 0);
                    pb[e] += sb;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;

        const int npro = min(nk, NS);
        for (int t = 0; t < npro; ++t) issue(t);
        retire(npro - 1);
        __builtin_amdgcn_s_barrier();
        static_for<NFR>([&](auto R) __attribute__((always_inline)) { rd1(I0{}, R, smem, smem + SA, 0); });

        int st_out = 0;
        // STEADY tiles (every tile but the last NS + 1) have no K tail, always issue tile kt + NS and always read tile kt + 1: their body is
        // ONE basic block, so the DMA issues and the next tile's first reads sit between the matrix instructions of k-group 3.
        auto tile = [&](auto steady_tag, int kt) __attribute__((always_inline)) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            const float* As = smem + st_out * STAGE;
            const float* Bs = As + SA;
            st_out = (st_out + 1 == NS) ? 0 : st_out + 1;
            const float* An = smem + st_out * STAGE;      // next tile's slot
            group(I0{}, std::false_type{}, As, Bs, 1, true, nullptr);
            group(I1{}, std::false_type{}, As, Bs, 2, true, nullptr);
            group(I0{}, std::false_type{}, As, Bs, 3, true, nullptr);
            if constexpr (STEADY) wait_vmcnt<(NS - 2) * NPT>();       // my pieces of tile kt + 1; tiles kt + 2 .. kt + NS - 1 stay in flight
            else if (kt + 1 < nk) retire(min(NS - 2, nk - 2 - kt));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my reads of (kt, group 3): the slot may be overwritten behind the barrier
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (STEADY) {
                float* Ad = smem + st_in * STAGE;
                st_in = (st_in + 1 == NS) ? 0 : st_in + 1;
                group(I1{}, std::true_type{}, An, An + SA, 0, true, Ad);
            } else {
                if (kt + NS < nk) issue(kt + NS);
                group(I1{}, std::false_type{}, An, An + SA, 0, kt + 1 < nk, nullptr);
            }
        };
        int kt = 0;
        for (; kt < nk - NS - 1; ++kt) tile(std::true_type{}, kt);
        for (; kt < nk; ++kt) tile(std::false_type{}, kt);
    }
};

template <int BM, int BN, bool A_KC, bool B_KC, int NS>
__global__ __launch_bounds__(256, 1) void gemm_sw_kernel(const GemmArgs g) {
    using L = SwLoop<BM, BN, A_KC, B_KC, NS>;
    constexpr int TM = L::TM, TN = L::TN, KB = L::KB;
    __shared__ __attribute__((aligned(16))) float smem[NS * L::STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const TileCoord tc = decode_tile(blockIdx.x, g.tiles_m, g.tiles_n, g.splits, g.split_map);
    const int m0 = tc.m * BM, n0 = tc.n * BN;
    const int kbeg = tc.split * g.kchunk;
    const int kend = min(g.Kloop, kbeg + g.kchunk);
    const int nk = (kend - kbeg) / KB;
    const bool tail_here = g.ktail && kend == g.Kloop;

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

    L::run(g, smem, m0, n0, kbeg, nk, tail_here, acc, asum, do_asum, wave, lane);
    if constexpr (!A_KC) {
        if (do_asum) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v = asum[i] + __shfl_xor(asum[i], 32, 64);
                const int row = m0 + wm0 + 32 * i + l31;
                if (half == 0 && row < g.M) {
                    if (g.splits > 1) g.asum_ws[(int64_t)tc.split * g.M + row] = v;
                    else g.asum[row] = v;
                }
            }
        }
    }
    __builtin_amdgcn_s_barrier();        // every wave is done reading the ring (and no DMA is in flight): it becomes the epilogue's staging space
    static_assert(4 * 64 * 32 * TN * 4 <= NS * L::STAGE * 4, "epilogue staging does not fit the ring");
    epilogue_lds<TM, TN>(g, acc, smem + wave * (64 * 32 * TN), m0 + wm0, n0 + wn0, lane, tc.split);
}

template <int BM, int BN, int NS>
static void launch_sw_tile(const GemmArgs& g, int transA, int transB, unsigned grid, hipStream_t s) {
    const dim3 gr(grid), blk(256);
    if (!transA && transB) hipLaunchKernelGGL((gemm_sw_kernel<BM, BN, true, true, NS>), gr, blk, 0, s, g);
    else if (!transA && !transB) hipLaunchKernelGGL((gemm_sw_kernel<BM, BN, true, false, NS>), gr, blk, 0, s, g);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_sw_kernel<BM, BN, false, false, NS>), gr, blk, 0, s, g);
    else hipLaunchKernelGGL((gemm_sw_kernel<BM, BN, false, true, NS>), gr, blk, 0, s, g);
}
bool launch_sw(int bm, int bn, const GemmArgs& g, int transA, int transB, unsigned grid, hipStream_t s) {
    if (bm == 256 && bn == 256) { launch_sw_tile<256, 256, 2>(g, transA, transB, grid, s); return true; }
    if (bm == 256 && bn == 128) { launch_sw_tile<256, 128, 3>(g, transA, transB, grid, s); return true; }
    return false;
}

}  // namespace ytvln
