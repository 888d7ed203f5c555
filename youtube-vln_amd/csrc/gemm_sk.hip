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
    int gshift;                 // log2(gsize) when gsize is a power of two, else -1
    int nk;                     // k-tiles per output tile
    int dp;                     // 1: whole tiles only (tile r, r + gsize, ... of the group), 0: stream-K ranges
    int tile_begin[9];          // group x owns tiles [tile_begin[x], tile_begin[x + 1]) of the grouped tile order
    float* partials;            // [2 G][slot]: slot = one accumulator dump in fragment order (NW * 64 lanes x 16 * TM * TN floats); first bank: partial tiles of
                                // the producers, second bank: the heads' own accumulators
    unsigned* ctl;              // zero-initialised: [0, 8) tickets, [8, 16) done counters, [16, 16 + G) partial-ready flags
    unsigned long long* probe;  // optional timeline: 16 s_memrealtime stamps (100 MHz) per workgroup
};

constexpr int SK_CTL_TICKET = 0, SK_CTL_DONE = 8, SK_CTL_FLAG = 16;

// range r of a group's iteration space [0, I = tiles x k-tiles): [floor(I r / gsize), floor(I (r + 1) / gsize)).  32-bit arithmetic (the host
// checks I * gsize < 2^31): a 64-bit division is a ~130-instruction VALU routine that made hipcc spill all 128 accumulators around it.
__device__ __forceinline__ void sk_range(const SkArgs& sk, int r, int ntg, int& s, int& e) {
    const unsigned I = (unsigned)ntg * (unsigned)sk.nk;
    if (sk.gshift >= 0) { s = (int)((I * (unsigned)r) >> sk.gshift); e = (int)((I * (unsigned)(r + 1)) >> sk.gshift); }
    else { s = (int)(I * (unsigned)r / (unsigned)sk.gsize); e = (int)(I * (unsigned)(r + 1) / (unsigned)sk.gsize); }
}

// 16-byte write-through store (sc1: visible to every XCD once acknowledged, no release fence needed): scalar base + per-lane byte offset
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(uint32_t voff, f32x4 v, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");      // (s_nop 4: see epi_store)
}

// four 16-byte loads (scalar bases + one per-lane offset) AND their wait in ONE statement: the destination registers are defined for the
// compiler only once the data has landed, whatever the register allocator does around the statement (the 128x128 form of this
// kernel has a scratch segment)
__device__ __forceinline__ void load16x4(f32x4 (&b)[4], uint32_t voff, const void* s0, const void* s1, const void* s2, const void* s3) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %6\n\tglobal_load_dwordx4 %2, %4, %7\n\t"
                 "global_load_dwordx4 %3, %4, %8\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]) : "v"(voff), "s"(s0), "s"(s1), "s"(s2), "s"(s3) : "memory");
}

template <int BM, int BN, bool A_KC, bool B_KC, int NW, int KB, bool SK, int WPS = 2>
__global__ __launch_bounds__(NW * 64, WPS) void gemm_sk_kernel(const GemmArgs g, const SkArgs sk) {
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
        if (probe && tid == 0 && pidx < 15) probe[pidx] = __builtin_amdgcn_s_memrealtime();
        ++pidx;
    };
    stamp();                                                              // [0] start

    // ---- place in the range order: stream-K draws a ticket (see the header), whole-tile launches wait for nobody and use the block id ------
    int r;
    if constexpr (!SK) {
        r = bid / sk.ngroups;
    } else {
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(sk.ctl + SK_CTL_TICKET + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            reinterpret_cast<volatile int*>(smem)[NS * STAGE] = (int)t;
        }
        __syncthreads();
        r = sk.gsize - 1 - __builtin_amdgcn_readfirstlane(reinterpret_cast<volatile int*>(smem)[NS * STAGE]);      // descending with the ticket
    }
    const int tb = sk.tile_begin[group], ntg = sk.tile_begin[group + 1] - tb;
    const int nk = sk.nk;
    const int my_slot = group * sk.gsize + r;
    stamp();                                                              // [1] range known

    int c_tile, c_k, c_rem, tile_step;
    if constexpr (!SK) {
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
    // a stream-K range that ends inside a tile owns that tile's HEAD: the partial tiles to add are those of ranges r + 1 .. jend - 1 (every range
    // is non-empty: the host checks it).  Worked out here, before the accumulators are live: any division or loop nest between the main loop and
    // the epilogue made hipcc spill all 128 of them.
    int jend = r + 1;
    if (SK && c_rem > 0) {
        int s, e;
        sk_range(sk, r, ntg, s, e);
        int kk = e % nk;                          // k-tiles of the last tile this range covers (0: it ends on a tile boundary)
        if (kk > 0 && e - s >= kk) {              // ... and it holds that tile's k = 0 end
            while (kk < nk) {
                int s2, e2;
                sk_range(sk, jend, ntg, s2, e2);
                kk += min(e2 - s2, nk - kk);
                ++jend;
            }
        }
    }
    int i_tile = c_tile, i_k = c_k, i_rem = c_rem;        // issue cursor: one k-tile ahead of the compute cursor, across piece boundaries

    const float* pa[TA::NI];
    const float* pb[TB::NI];
    const int64_t sa = TA::step(g.lda), sb = TB::step(g.ldb);
    int st_in = 0, st_out = 0;

    // per-lane source pointers of the k-tile the issue cursor stands on (rebuilt from scalars at every tile change AND behind every epilogue, so
    // that no address register has to survive the epilogue next to the accumulators)
    auto setup_ptrs = [&]() {
        const TileCoord tc = tile_coord(tb + i_tile, g.tiles_m, g.tiles_n);
#pragma unroll
        for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, tc.m * BM, i_k * KB, wave, lane, i, 0x7fffffff);
#pragma unroll
        for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, tc.n * BN, i_k * KB, wave, lane, i, 0x7fffffff);
    };
    auto issue = [&]() {
        float* As = smem + st_in * STAGE;
        float* Bs = As + SA;
        st_in ^= 1;
        if (g.ktail && i_k == nk - 1) {          // K tail: clamped k rows (the tile ends with this k-tile: the pointers are rebuilt behind it)
            const TileCoord tc = tile_coord(tb + i_tile, g.tiles_m, g.tiles_n);
#pragma unroll
            for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, tc.m * BM, i_k * KB, wave, lane, i, g.K - 1);
#pragma unroll
            for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, tc.n * BN, i_k * KB, wave, lane, i, g.K - 1);
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
        --i_rem;
        if (++i_k == nk) {
            i_k = 0; i_tile += tile_step;
            if (i_rem > 0) setup_ptrs();
        }
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

    if (i_rem > 0) { setup_ptrs(); issue(); }
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
            if (!SK || (c_k0 == 0 && c_k == nk)) {
                gemm_epilogue<TM, TN, BM != 128>(g, acc, tc.m * BM + wm0, tc.n * BN + wn0, l31, half, 0);
            } else if constexpr (SK) {
                // ---- a piece of a shared tile: dump the accumulators in fragment order (write-through: visible on every XCD once acknowledged) ------
                // Both the producers (k0 > 0) and the head (k0 == 0) do this: the head then rebuilds the tile from memory, 32x32 sub-tile by
                // sub-tile, so the 128 accumulator registers are only ever READ here (a path that modifies them in front of the epilogue makes
                // hipcc spill all of them around every piece end, whole tiles included).
                const uint32_t voff = (uint32_t)tid * 16u;
                {
                    // (a head parks its own accumulators in the second bank of slots: its first piece may have published slot my_slot already)
                    const char* const dst = reinterpret_cast<const char*>(sk.partials + (size_t)(my_slot + (c_k0 > 0 ? 0 : sk.G)) * SLOT);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                store16_sc1(voff, f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]},
                                            dst + (size_t)((i * TN + j) * 4 + q) * (NW * 64 * 16));
                }
                wait_vmcnt<0>();                  // every storing wave drains its write-through stores
                __syncthreads();
                if (c_k0 > 0) {                   // producer: raise the flag
                    if (tid == 0) __hip_atomic_store(sk.ctl + SK_CTL_FLAG + my_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {                          // head: wait for ranges r + 1 .. jend - 1 (one wave polls relaxed, one acquire), then add in ascending k order
                    if (wave == 0) {
                        for (int j = r + 1; j < jend; ++j)
                            while (__hip_atomic_load(sk.ctl + SK_CTL_FLAG + group * sk.gsize + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                                __builtin_amdgcn_s_sleep(4);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const char* const base = reinterpret_cast<const char*>(sk.partials + (size_t)(group * sk.gsize) * SLOT);
#pragma unroll 1
                    for (int gi = 0; gi < TM * TN; ++gi) {
                        f32x16 sum[1][1];
                        {
                            f32x4 buf[4];         // (a thread reads back exactly the 16-byte pieces it wrote itself)
                            {
                                const char* const b0 = base + (size_t)(sk.G + r) * (SLOT * 4) + (size_t)(gi * 4) * (NW * 64 * 16);
                                load16x4(buf, voff, b0, b0 + (size_t)(NW * 64 * 16), b0 + (size_t)2 * (NW * 64 * 16), b0 + (size_t)3 * (NW * 64 * 16));
                            }
                            typedef float f32x8 __attribute__((ext_vector_type(8)));
                            const f32x8 lo = __builtin_shufflevector(buf[0], buf[1], 0, 1, 2, 3, 4, 5, 6, 7);
                            const f32x8 hi = __builtin_shufflevector(buf[2], buf[3], 0, 1, 2, 3, 4, 5, 6, 7);
                            sum[0][0] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                        }
#pragma unroll 1
                        for (int j = r + 1; j < jend; ++j) {
                            f32x4 buf[4];
                            {
                                const char* const b0 = base + (size_t)j * (SLOT * 4) + (size_t)(gi * 4) * (NW * 64 * 16);
                                load16x4(buf, voff, b0, b0 + (size_t)(NW * 64 * 16), b0 + (size_t)2 * (NW * 64 * 16), b0 + (size_t)3 * (NW * 64 * 16));
                            }
                            typedef float f32x8 __attribute__((ext_vector_type(8)));
                            const f32x8 lo = __builtin_shufflevector(buf[0], buf[1], 0, 1, 2, 3, 4, 5, 6, 7);
                            const f32x8 hi = __builtin_shufflevector(buf[2], buf[3], 0, 1, 2, 3, 4, 5, 6, 7);
                            sum[0][0] += __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                        }
                        gemm_epilogue<1, 1, BM != 128>(g, sum, tc.m * BM + wm0 + 32 * (gi / TN), tc.n * BN + wn0 + 32 * (gi % TN), l31, half, 0);
                    }
                    if (tid == 0)
                        for (int j = r + 1; j < jend; ++j)
                            __hip_atomic_store(sk.ctl + SK_CTL_FLAG + group * sk.gsize + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            stamp();                              // epilogue / dump issued
            zero_acc();
            c_k0 = 0;
            if (c_k == nk) { c_k = 0; c_tile += tile_step; }
            if (i_rem > 0) setup_ptrs();          // (the next k-tile of the issue cursor: see setup_ptrs)
        }
    }
    // ---- stream-K: leave the control block zeroed -- the last workgroup of the group to finish resets ticket and done counters -------------
    if (SK && tid == 0) {
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

// tile codes as in gemm.hip's planner: 0 = 128x128 (8 waves of 32x64, TWO workgroups per CU: 32 accumulator registers per lane, 64 KB of LDS),
// 3 = 256x128, 4 = 256x256 (8 waves, one workgroup per CU)
static int sk_bm(int tile) { return tile == 0 ? 128 : 256; }
static int sk_bn(int tile) { return tile == 4 ? 256 : 128; }
static int sk_per_cu(int tile) { return tile == 0 ? 2 : 1; }
// us per 32-deep k-tile of ONE workgroup with the CU fully occupied (round-5 timelines: 256x256 7.7-8.3 in the persistent kernel against 7.5 in the
// launch-per-tile one -- scalar-register pressure puts lane reads of spilled scalars into its main loop; 256x128 4.5; 128x128: 2.06 per CU = 4.12 per
// workgroup when two share it)
static double sk_tk(int tile) { return tile == 4 ? 8.0 : tile == 3 ? 4.5 : 4.12; }

SkPlan plan_sk(int M, int N, int Kloop, int transA, int epilogue, bool fast, bool x3) {
    SkPlan p = {0, 4, 1, 0, 8, 0.0};
    const int mode = opt(OPT_GEMM_SK);              // 0 off, 1 planner, 2 force DP, 3 force SK
    if (!mode || !fast || x3 || transA || M < 256 || N < 128 || Kloop % BK) return p;
    const int nk = Kloop / BK;
    const int ngroups = opt(OPT_GEMM_SK_GROUPS) == 1 ? 1 : 8;
    const int force_tile = opt(OPT_GEMM_SK_TILE);
    double best = 1e30;
    static const int tiles[3] = {0, 3, 4};
    for (int ti = 0; ti < 3; ++ti) {
        const int tile = tiles[ti];
        if (force_tile >= 0 && tile != force_tile) continue;
        const int bm = sk_bm(tile), bn = sk_bn(tile);
        if (N < bn) continue;
        const int G = sk_num_cus() * sk_per_cu(tile);
        const int ntiles = (int)(cdiv(M, bm) * cdiv(N, bn));
        if (ntiles < 2 * ngroups) continue;
        const int gsize = G / ngroups;
        const int ntg = (int)cdiv(ntiles, ngroups);                    // the fullest group
        for (int dp = 0; dp <= 1; ++dp) {
            if ((mode == 2 && !dp) || (mode == 3 && dp)) continue;
            const double iters = dp ? (double)cdiv(ntg, gsize) * nk : (double)cdiv((int64_t)ntg * nk, gsize);
            if (!dp && (iters < 4 || (double)ntg * nk * gsize >= 2147483648.0 || (int64_t)(ntiles / ngroups) * nk < gsize)) continue;
            const double tile_bytes = (double)bm * bn * 4.0;
            // DP: ~2 us of un-overlapped epilogue per tile; SK: one partial dump and, for the heads, the own tile + (pieces - 1) partial reads at ~100 GB/s
            const double pieces = dp ? 1.0 : std::max(1.0, (double)nk / iters + 1.0);
            double t = 6.0 + iters * sk_tk(tile) + (dp ? 2.0 * cdiv(ntg, gsize) : 3.0 + tile_bytes / 100e3 * (pieces + 1.0));
            if ((epilogue != YTVLN_EPI_NONE)) t += 3.0 * (dp ? (double)cdiv(ntg, gsize) : 1.0);
            if (t < best) { best = t; p.use = 1; p.tile = tile; p.dp = dp; p.G = G; p.ngroups = ngroups; p.cost = t; }
        }
    }
    return p;
}

int64_t sk_workspace_elems(const SkPlan& p) {
    if (!p.use || p.dp) return 0;
    return (int64_t)2 * p.G * sk_bm(p.tile) * sk_bn(p.tile);
}

template <int BM, int BN, int WPS>
static void sk_launch_tile(const GemmArgs& g, const SkArgs& sk, int transB, hipStream_t s) {
    const dim3 grid(sk.G), blk(512);
    if (sk.dp) {
        if (transB) hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, true, true, 8, 32, false, WPS>), grid, blk, 0, s, g, sk);
        else hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, true, false, 8, 32, false, WPS>), grid, blk, 0, s, g, sk);
    } else {
        if (transB) hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, true, true, 8, 32, true, WPS>), grid, blk, 0, s, g, sk);
        else hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, true, false, 8, 32, true, WPS>), grid, blk, 0, s, g, sk);
    }
}

void sk_launch(GemmArgs& g, const SkPlan& p, int transB, float* partials, unsigned* ctl, hipStream_t s) {
    const int bm = sk_bm(p.tile), bn = sk_bn(p.tile);
    g.tiles_m = (int)cdiv(g.M, bm);
    g.tiles_n = (int)cdiv(g.N, bn);
    g.ntiles = g.tiles_m * g.tiles_n;
    g.splits = 1;
    SkArgs sk;
    sk.G = p.G; sk.ngroups = p.ngroups; sk.gsize = p.G / p.ngroups;
    sk.gshift = -1;
    for (int b = 0; b < 12; ++b) if ((1 << b) == sk.gsize) sk.gshift = b;
    sk.nk = g.Kloop / BK;
    sk.dp = p.dp;
    for (int x = 0; x <= 8; ++x) sk.tile_begin[x] = x <= p.ngroups ? (int)((int64_t)g.ntiles * x / p.ngroups) : g.ntiles;
    sk.partials = partials;
    sk.ctl = ctl;
    sk.probe = g_probe;
    if (p.tile == 4) sk_launch_tile<256, 256, 2>(g, sk, transB, s);
    else if (p.tile == 3) sk_launch_tile<256, 128, 2>(g, sk, transB, s);
    else sk_launch_tile<128, 128, 4>(g, sk, transB, s);
}

}  // namespace ytvln

extern "C" int64_t ytvln_gemm_sk_ctl_elems(void) { return ytvln::SK_CTL_FLAG + 1024; }

extern "C" int ytvln_gemm_probe(unsigned long long* buffer) {
    ytvln::g_probe = buffer;
    return 0;
}
