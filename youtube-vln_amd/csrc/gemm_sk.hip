// Persistent fp32 GEMM on the CDNA4 matrix cores: one workgroup per CU walks a list of k-tile "iterations" of the output tiles.
//
// Why (round 5): the launch-per-tile kernel of gemm.hip pays ramp + first-tile latency + the epilogue's store burst un-overlapped for
// every 256x256 tile (one workgroup per CU: the next workgroup cannot start before the previous one has released its 128 KB of LDS), and
// launches whose tile count is not a multiple of the 256 CUs leave CUs idle for a whole tile (the 4480-row text shapes: 210-216 tiles).
// Here
//   * the k-tiles of consecutive pieces form ONE stream through the two-slot LDS ring: the first k-tile of the next piece is in flight
//     (LDS-DMA) while the accumulators of the finished piece drain through the epilogue;
//   * "DP" launches give every workgroup whole tiles (tile r, r + gsize, ...), "SK" (stream-K) launches cut the iteration space of a
//     ticket group -- tiles x k-tiles, tile-major -- into equal contiguous ranges, so a tile may be shared by several workgroups:
//     the workgroup that holds the k = 0 end of a tile (the HEAD, which it reaches at the END of its range) adds the partial tiles the
//     later ranges produced at the START of theirs, in ascending k order (fixed order -> bit-reproducible), and runs the fused epilogue.
//   * Deadlock freedom without any co-residency assumption: a workgroup's place in the range order is a TICKET drawn at start-up
//     (atomic counter per group), ranges are handed out in DESCENDING order, so the producer of every partial a head waits for holds a
//     lower ticket = has already started.  (Two persistent launches on two streams may interleave on the CUs: waiting on a workgroup
//     that is not resident yet is what must never happen.)  The control block (tickets, done counters, flags) is zero when a launch
//     starts and is left zero by it: the last workgroup of a group to finish resets the counters, a head resets the flags it consumed.
//   * ticket groups = XCDs (blockIdx % 8 with grid = 256): group x owns a contiguous run of whole tiles of the "grouped" tile order
//     (8 tile rows x a few columns: the panels its 32 workgroups stream meet in that XCD's L2), so partial tiles never cross an XCD.
#include "gemm_tiles.h"

namespace ytvln {

struct SkArgs {
    int G, ngroups, gsize;      // workgroups in the launch, ticket groups, workgroups per group
    int nk;                     // k-tiles per output tile
    int dp;                     // 1: whole tiles only (tile r, r + gsize, ... of the group), 0: stream-K ranges
    int krot;                   // k rotation: tile t starts its contraction at k-tile (t * krot) % nk (spreads the operand rows the workgroups of a
                                // launch read at the same time over the memory channels); 0 = every tile starts at k = 0
    int tile_begin[9];          // group x owns tiles [tile_begin[x], tile_begin[x + 1]) of the grouped tile order
    float* partials;            // [G][slot]: slot = one accumulator dump in fragment order (NW * 64 lanes x 16 * TM * TN floats)
    unsigned* ctl;              // zero-initialised: [0, 8) tickets, [8, 16) done counters, [16, 16 + G) partial-ready flags
    unsigned long long* probe;  // optional timeline: 16 s_memrealtime stamps (100 MHz) per workgroup
};

constexpr int SK_CTL_TICKET = 0, SK_CTL_DONE = 8, SK_CTL_FLAG = 16;

__device__ __forceinline__ void sk_range(const SkArgs& sk, int r, int ntg, int& s, int& e) {
    const long long I = (long long)ntg * sk.nk;
    s = (int)(I * r / sk.gsize);
    e = (int)(I * (r + 1) / sk.gsize);
}

template <int BM, int BN, bool A_KC, bool B_KC, int NW, int KB>
__global__ __launch_bounds__(NW * 64, 2) void gemm_sk_kernel(const GemmArgs g, const SkArgs sk) {
    using TA = DmaTile<BM, A_KC, NW, KB>;
    using TB = DmaTile<BN, B_KC, NW, KB>;
    constexpr int NS = 2;
    constexpr int WM = NW / 2;
    constexpr int TM = BM / WM / 32, TN = BN / 64;
    constexpr int SA = BM * KB, SB = BN * KB, STAGE = SA + SB;
    constexpr int NG = KB / 8;
    constexpr int NQ = TM * TN * 4;                       // float4 per lane in one accumulator dump
    constexpr int SLOT = NW * 64 * NQ * 4;                // floats per partial slot
    __shared__ __attribute__((aligned(16))) float smem[NS * STAGE + 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * (BM / WM), wn0 = (wave & 1) * (BN / 2);
    const int bid = blockIdx.x;
    const int group = bid % sk.ngroups;
    unsigned long long* probe = sk.probe ? sk.probe + (size_t)bid * 16 : nullptr;
    int pidx = 0;
    auto stamp = [&]() {
        if (probe && tid == 0 && pidx < 16) probe[pidx] = __builtin_amdgcn_s_memrealtime();
        ++pidx;
    };
    stamp();                                                              // [0] start

    // ---- ticket -> range ------------------------------------------------------------------------------------------------------------
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(sk.ctl + SK_CTL_TICKET + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<volatile int*>(smem)[NS * STAGE] = (int)t;
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(reinterpret_cast<volatile int*>(smem)[NS * STAGE]);
    const int r = sk.gsize - 1 - ticket;                                  // range index inside the group (descending with the ticket)
    const int tb = sk.tile_begin[group], ntg = sk.tile_begin[group + 1] - tb;
    const int nk = sk.nk;
    const int my_slot = group * sk.gsize + r;
    stamp();                                                              // [1] ticket drawn

    int c_tile, c_k, c_rem, tile_step;
    if (sk.dp) {
        tile_step = sk.gsize;
        c_tile = r; c_k = 0;
        c_rem = r < ntg ? ((ntg - r + sk.gsize - 1) / sk.gsize) * nk : 0;
    } else {
        int s, e;
        sk_range(sk, r, ntg, s, e);
        tile_step = 1;
        c_tile = s / nk; c_k = s - c_tile * nk; c_rem = e - s;
    }
    int c_k0 = c_k;
    int i_tile = c_tile, i_k = c_k, i_rem = c_rem;
    bool i_new = true;

    const float* pa[TA::NI];
    const float* pb[TB::NI];
    const int64_t sa = TA::step(g.lda), sb = TB::step(g.ldb);
    int kp = 0;                                                           // physical k-tile of the next issue
    int st_in = 0, st_out = 0;

    auto issue = [&]() {
        if (i_new) {
            const TileCoord tc = tile_coord(tb + i_tile, g.tiles_m, g.tiles_n);
            kp = i_k + (sk.krot ? (int)(((long long)(tb + i_tile) * sk.krot) % nk) : 0);
            if (kp >= nk) kp -= nk;
#pragma unroll
            for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, tc.m * BM, kp * KB, wave, lane, i, 0x7fffffff);
#pragma unroll
            for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, tc.n * BN, kp * KB, wave, lane, i, 0x7fffffff);
            i_new = false;
        }
        float* As = smem + st_in * STAGE;
        float* Bs = As + SA;
        st_in ^= 1;
        if (g.ktail && kp == nk - 1) {           // K tail (krot == 0 there): clamped k rows; the piece ends with this k-tile, the pointers are rebuilt
            const TileCoord tc = tile_coord(tb + i_tile, g.tiles_m, g.tiles_n);
#pragma unroll
            for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, tc.m * BM, kp * KB, wave, lane, i, g.K - 1);
#pragma unroll
            for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, tc.n * BN, kp * KB, wave, lane, i, g.K - 1);
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
        if (++kp == nk) {                        // rotated contraction wraps to the first k-tile
            kp = 0;
#pragma unroll
            for (int i = 0; i < TA::NI; ++i) pa[i] -= (int64_t)nk * sa;
#pragma unroll
            for (int i = 0; i < TB::NI; ++i) pb[i] -= (int64_t)nk * sb;
        }
        --i_rem;
        if (++i_k == nk) { i_k = 0; i_tile += tile_step; i_new = true; }
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    };
    zero_acc();

    if (i_rem > 0) issue();
    bool first = true;
    while (c_rem > 0) {
        wait_vmcnt<0>();                          // my pieces of this k-tile have landed (and the previous piece's stores are acknowledged)
        __builtin_amdgcn_s_barrier();             // everybody's have; everybody is done reading the other slot
        if (first) { stamp(); first = false; }    // [2] first operands in LDS
        if (i_rem > 0) issue();
        const float* As = smem + st_out * STAGE;
        const float* Bs = As + SA;
        st_out ^= 1;
#pragma unroll
        for (int sg = 0; sg < NG; ++sg) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = TA::frag(As, wm0, i, l31, half, sg);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = TB::frag(Bs, wn0, j, l31, half, sg);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        ++c_k; --c_rem;
        if (c_k == nk || c_rem == 0) {            // the piece [c_k0, c_k) of tile c_tile is complete
            stamp();                              // main loop of the piece done
            const TileCoord tc = tile_coord(tb + c_tile, g.tiles_m, g.tiles_n);
            if (c_k0 > 0) {
                // ---- producer: dump the accumulators in fragment order, then raise the flag -----------------------------------------------
                float4* dst = reinterpret_cast<float4*>(sk.partials + (size_t)my_slot * SLOT) + tid;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            dst[(size_t)((i * TN + j) * 4 + q) * (NW * 64)] =
                                make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                wait_vmcnt<0>();                  // acknowledged by the L2
                __syncthreads();
                if (tid == 0) __hip_atomic_store(sk.ctl + SK_CTL_FLAG + my_slot, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (c_k < nk) {
                    // ---- head: add the partial tiles of the following ranges in ascending k order ------------------------------------------
                    int kk = c_k, j = r + 1;
                    while (kk < nk) {
                        int s, e;
                        sk_range(sk, j, ntg, s, e);
                        if (e == s) { ++j; continue; }          // an empty range (fewer iterations than workgroups) produces nothing
                        const int slot = group * sk.gsize + j;
                        while (__hip_atomic_load(sk.ctl + SK_CTL_FLAG + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                            __builtin_amdgcn_s_sleep(2);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        const float4* src = reinterpret_cast<const float4*>(sk.partials + (size_t)slot * SLOT) + tid;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float4 v = src[(size_t)((i * TN + jj) * 4 + q) * (NW * 64)];
                                    acc[i][jj][4 * q] += v.x; acc[i][jj][4 * q + 1] += v.y;
                                    acc[i][jj][4 * q + 2] += v.z; acc[i][jj][4 * q + 3] += v.w;
                                }
                        kk += min(e - s, nk - kk);
                        ++j;
                    }
                    __syncthreads();              // every wave has seen the flags: clear them for the next launch
                    if (tid == 0)
                        for (int jj = r + 1; jj < j; ++jj)
                            __hip_atomic_store(sk.ctl + SK_CTL_FLAG + group * sk.gsize + jj, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                gemm_epilogue<TM, TN>(g, acc, tc.m * BM + wm0, tc.n * BN + wn0, l31, half, 0);
            }
            stamp();                              // epilogue / dump issued
            zero_acc();
            c_k0 = 0;
            if (c_k == nk) { c_k = 0; c_tile += tile_step; }
        }
    }
    // ---- leave the control block zeroed: the last workgroup of the group to finish resets ticket and done counters --------------------
    if (tid == 0) {
        const unsigned d = __hip_atomic_fetch_add(sk.ctl + SK_CTL_DONE + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)d == sk.gsize - 1) {
            __hip_atomic_store(sk.ctl + SK_CTL_TICKET + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sk.ctl + SK_CTL_DONE + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (probe && tid == 0) probe[15] = __builtin_amdgcn_s_memrealtime();     // end
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
static unsigned long long* g_probe = nullptr;

int sk_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8) n = v / 8 * 8;
        else n = 256;
    }
    return n;
}

// us per 32-deep k-tile at the CU-exclusive rate (gemm.hip plan_cost: 256x256 8.0, 256x128 4.16) and the fixed costs of a persistent launch
static double sk_tk(int tile) { return tile == 4 ? 8.0 : 4.16; }

SkPlan plan_sk(int M, int N, int Kloop, int transA, int epilogue, bool fast, bool x3) {
    SkPlan p = {0, 4, 1, 0, 8, 0.0};
    const int mode = opt(OPT_GEMM_SK);              // 0 off, 1 planner, 2 force DP, 3 force SK
    if (!mode || !fast || x3 || transA || M < 256 || N < 128 || Kloop % BK) return p;
    const int G = sk_num_cus(), nk = Kloop / BK;
    const int ngroups = opt(OPT_GEMM_SK_GROUPS) == 1 ? 1 : 8;
    const int force_tile = opt(OPT_GEMM_SK_TILE);
    double best = 1e30;
    for (int tile = 3; tile <= 4; ++tile) {
        if (force_tile >= 0 && tile != force_tile) continue;
        const int bm = 256, bn = tile == 4 ? 256 : 128;
        if (N < bn) continue;
        const int ntiles = (int)(cdiv(M, bm) * cdiv(N, bn));
        if (ntiles < 2 * ngroups) continue;
        const int gsize = G / ngroups;
        const int ntg = (int)cdiv(ntiles, ngroups);                    // the fullest group
        for (int dp = 0; dp <= 1; ++dp) {
            if ((mode == 2 && !dp) || (mode == 3 && dp)) continue;
            const double iters = dp ? (double)cdiv(ntg, gsize) * nk : (double)cdiv((int64_t)ntg * nk, gsize);
            if (!dp && iters < 4) continue;
            const double tile_bytes = (double)bm * bn * 4.0;
            // DP: ~2 us of un-overlapped epilogue per tile; SK: one partial dump and, for the heads, (pieces - 1) partial reads at ~150 GB/s
            const double pieces = dp ? 1.0 : std::max(1.0, (double)nk / iters + 1.0);
            double t = 6.0 + iters * sk_tk(tile) + (dp ? 2.0 * cdiv(ntg, gsize) : 3.0 + tile_bytes / 150e3 * pieces);
            if ((epilogue != YTVLN_EPI_NONE)) t += 3.0 * (dp ? (double)cdiv(ntg, gsize) : 1.0);
            if (t < best) { best = t; p.use = 1; p.tile = tile; p.dp = dp; p.G = G; p.ngroups = ngroups; p.cost = t; }
        }
    }
    return p;
}

int64_t sk_workspace_elems(const SkPlan& p) {
    if (!p.use || p.dp) return 0;
    const int bn = p.tile == 4 ? 256 : 128;
    return (int64_t)p.G * 256 * bn;
}

template <int BM, int BN>
static void sk_launch_tile(const GemmArgs& g, const SkArgs& sk, int transB, hipStream_t s) {
    const dim3 grid(sk.G), blk(512);
    if (transB) hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, true, true, 8, 32>), grid, blk, 0, s, g, sk);
    else hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, true, false, 8, 32>), grid, blk, 0, s, g, sk);
}

void sk_launch(GemmArgs& g, const SkPlan& p, int transB, float* partials, unsigned* ctl, hipStream_t s) {
    const int bm = 256, bn = p.tile == 4 ? 256 : 128;
    g.tiles_m = (int)cdiv(g.M, bm);
    g.tiles_n = (int)cdiv(g.N, bn);
    g.ntiles = g.tiles_m * g.tiles_n;
    g.splits = 1;
    SkArgs sk;
    sk.G = p.G; sk.ngroups = p.ngroups; sk.gsize = p.G / p.ngroups;
    sk.nk = g.Kloop / BK;
    sk.dp = p.dp;
    sk.krot = g.ktail ? 0 : std::max(0, opt(OPT_GEMM_KROT));
    for (int x = 0; x <= 8; ++x) sk.tile_begin[x] = x <= p.ngroups ? (int)((int64_t)g.ntiles * x / p.ngroups) : g.ntiles;
    sk.partials = partials;
    sk.ctl = ctl;
    sk.probe = g_probe;
    if (p.tile == 4) sk_launch_tile<256, 256>(g, sk, transB, s);
    else sk_launch_tile<256, 128>(g, sk, transB, s);
}

}  // namespace ytvln

extern "C" int64_t ytvln_gemm_sk_ctl_elems(void) { return ytvln::SK_CTL_FLAG + 1024; }

extern "C" int ytvln_gemm_probe(unsigned long long* buffer) {
    ytvln::g_probe = buffer;
    return 0;
}
