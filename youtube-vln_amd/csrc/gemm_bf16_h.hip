// bf16 GEMM, main loop with 32-deep k-tiles in FIVE-slot LDS rings and three tiles in flight (run-time option GEMM_BF16_FORM = 3).
//
// Same arithmetic, operand layouts, wave grid (8 waves of 64x128 on a 256x256 tile; 32x64 on 128x128), epilogues and split-K as
// gemm_bf16_kernel (gemm_bf16.hip) -- what changes is how far ahead of its use an operand tile is requested.  There the 160 KB of LDS hold
// 64-deep tiles, A in three slots (requested two k-tiles ahead) and B in two (ONE k-tile ahead): B(kt+1) is issued in load phase L01(kt) and
// every wave has to see its pieces landed two phases later, in L23(kt) (the staggered wave group reads one barrier behind, so the wait cannot
// sit later) -- 0.8 us after the issue, while an LDS-DMA piece takes ~1.1 us from issue to landed once every CU streams (guide: ldsdma-fill).
// The counted vmcnt in front of that barrier therefore stalls in EVERY k-tile; moving the issue around inside the k-tile (forms 1, 2) cannot
// help and measured 0-4 % slower (profiles/round6_gemm_bf16_forms.log).  Here a tile is 32 deep (16 KB per operand), each operand has a ring
// of five slots and tile t+4 is requested while tile t is being read: three whole tiles (2.4 us of main loop) are in flight behind the one the
// wait names, for BOTH operands, in the same 160 KB.  Two phases per tile, each closed by a workgroup barrier:
//     L(t)  12 LDS fragment reads of tile t; issue the LDS-DMA of tile t+4 (A then B); wait until this wave's pieces of tile t+1 have landed
//           (vmcnt(3 x pieces per tile): tiles t+2 .. t+4 may still be in flight)
//     M(t)  16 matrix instructions (2 k-steps x TM x TN)
// and wave group 1 (waves 4-7) runs one barrier behind group 0, as in gemm_bf16_kernel.  Hazards (slot = interval between two barrier
// releases; group g runs phase p in slot p + g; L(t) is phase 2t):
//   * tile t+4 goes into the ring slot tile t-1 was read from; its last reader is group 1 in L(t-1) = slot 2t-1, retired (lgkmcnt(0)) before
//     the barrier that closes that slot; the earliest issue is group 0 in slot 2t;
//   * tile t+1 is first read by group 0 in slot 2t+2; every wave has waited for its own pieces of it in L(t) (slots 2t, 2t+1) and passed the
//     barrier closing slot 2t+1.  Vector memory retires in order and a tile's pieces are issued A first, B second, tile after tile, so
//     "at most 3 tiles' worth outstanding" = everything up to tile t+1 complete.
// LDS images (lane-linear 1 KiB DMA pieces, layout chosen through the per-lane SOURCE address):
//   contraction-contiguous operand  S[m][32]   64-byte rows, 16-byte granule g stored at g ^ ((m >> 2) & 3): the 16 rows of a ds_read_b128
//                                              lane group fall on 16 distinct 16-byte slots of the 256-byte bank row
//   k-major operand                 T[k][BMN]  as in gemm_bf16.hip (granule g of row k at g ^ (4 (k & 3))), gathered by ds_read_b64_tr_b16
// A lane's k slots inside a 32-deep tile are k = 16 h + 8 s + e (h = half-wave, s = matrix step 0..1, e = 0..7) for both layouts.
#define YT_BF16_SHARED_ONLY 1
#include "gemm_bf16.hip"

namespace ytvln {

constexpr int HK = 32;           // k-tile depth in bf16 elements
constexpr int HSLOTS = 5;        // ring slots per operand
constexpr int HAHEAD = 4;        // tile t+4 is requested while tile t is read

template <int BMN, bool KC, int NW>
struct HTile {
    static constexpr int PIECES = BMN / 16, NI = PIECES / NW;          // BMN x 32 bf16 = BMN / 16 pieces of 1 KiB
    static_assert(NI >= 1 && PIECES % NW == 0, "tile too small for the workgroup");
    static constexpr int ROWB = KC ? 64 : BMN * 2;
    static constexpr int RPP = 1024 / ROWB;                            // LDS rows per piece (KC: 16 rows of m; k-major: 2 or 4 rows of k)
    static constexpr int GPR = ROWB / 16;
    __device__ static __forceinline__ void coord(int c, int lane, int& row, int& lg) {
        row = c * RPP + lane / GPR;
        const int pg = lane % GPR;
        lg = KC ? (pg ^ ((row >> 2) & 3)) : (pg ^ (4 * (row & 3)));
    }
    __device__ static __forceinline__ const bf16_t* src(const bf16_t* P, int64_t ld, int MN, int mn0, int k0, int c, int lane) {
        int row, lg;
        coord(c, lane, row, lg);
        if (KC) return P + (int64_t)min(mn0 + row, MN - 1) * ld + k0 + 8 * lg;
        return P + (int64_t)(k0 + row) * ld + min(mn0 + 8 * lg, MN - 8);
    }
    __device__ static __forceinline__ int64_t step(int64_t ld) { return KC ? HK : HK * ld; }
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

template <int BMN, bool KC, int NSUB>
struct HFrag {
    int off[NSUB];
    int key[NSUB];          // KC only
    __device__ __forceinline__ void init(int w0, int lane) {
        const int l31 = lane & 31, h = lane >> 5;
        if (KC) {
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                const int row = w0 + 32 * i + l31;
                off[i] = row * 64;
                key[i] = (row >> 2) & 3;
            }
        } else {
            const int r = (lane & 15) >> 2, c4 = lane & 3, blk = (lane >> 4) & 1;
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                const int col = w0 + 32 * i + 16 * blk + 4 * c4;
                off[i] = (16 * h + r) * (BMN * 2) + (((col >> 3) ^ (4 * r)) << 4) + ((col & 7) << 1);
                key[i] = 0;
            }
        }
    }
    // the 8 bf16 this lane feeds to matrix step s (0 or 1) of the tile at S (sub-tile i)
    __device__ __forceinline__ bf16x8 get(const char* __restrict__ S, int i, int h, int s) const {
        if (KC) {
            return *reinterpret_cast<const bf16x8*>(S + off[i] + (((2 * h + s) ^ key[i]) << 4));
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

template <int N>
__device__ __forceinline__ void h_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int BM, int BN, bool A_KC, bool B_KC, int NW, int WPS, typename CT>
__global__ __launch_bounds__(NW * 64, WPS) void gemm_bf16_h_kernel(const BfArgs g) {
    using TA = HTile<BM, A_KC, NW>;
    using TB = HTile<BN, B_KC, NW>;
    constexpr int WM = NW / 2;
    constexpr int TM = BM / WM / 32, TN = BN / 64;
    constexpr int SA = BM * HK * 2, SB = BN * HK * 2;
    constexpr int NP = TA::NI + TB::NI;                                   // LDS-DMA pieces a wave issues per tile
    static_assert(3 * NP <= 63, "vmcnt field");
    __shared__ __attribute__((aligned(16))) char smem[HSLOTS * (SA + SB)];     // ONE shared object (a second one de-pipelines the DMA)
    char* const ringA = smem;
    char* const ringB = smem + HSLOTS * SA;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * (BM / WM), wn0 = (wave & 1) * (BN / 2);
    const BfCoord tc = bf_decode(blockIdx.x, g.tiles_m, g.tiles_n, g.splits);
    const int m0 = tc.m * BM, n0 = tc.n * BN;
    const int kbeg = tc.split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + HK - 1) / HK;
    const int nfull = (kend - kbeg) / HK;                     // whole tiles: LDS-DMA; a last partial one goes through registers

    const bf16_t* pa[TA::NI];
    const bf16_t* pb[TB::NI];
#pragma unroll
    for (int i = 0; i < TA::NI; ++i) pa[i] = TA::src(g.A, g.lda, g.mnA, m0, kbeg, wave * TA::NI + i, lane);
#pragma unroll
    for (int i = 0; i < TB::NI; ++i) pb[i] = TB::src(g.B, g.ldb, g.mnB, n0, kbeg, wave * TB::NI + i, lane);
    const int64_t sa = TA::step(g.lda), sb = TB::step(g.ldb);

    HFrag<BM, A_KC, TM> fa;
    HFrag<BN, B_KC, TN> fb;
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

    int slot_in = 0;                          // ring slot the next issue fills (the same index in both rings)
    auto issue = [&](int t) {
        char* As = ringA + slot_in * SA;
        char* Bs = ringB + slot_in * SB;
        slot_in = slot_in + 1 == HSLOTS ? 0 : slot_in + 1;
        if (t < nfull) {
#pragma unroll
            for (int i = 0; i < TA::NI; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)pa[i], (lds_ptr_t)(As + (wave * TA::NI + i) * 1024), 16, 0, 0);
                pa[i] += sa;
            }
#pragma unroll
            for (int i = 0; i < TB::NI; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)pb[i], (lds_ptr_t)(Bs + (wave * TB::NI + i) * 1024), 16, 0, 0);
                pb[i] += sb;
            }
        } else {          // the last tile crosses K: the same pieces through registers, zero past the end (the compiler waits for these loads itself
                          // -- which drains the DMA queue too -- and the lgkmcnt(0) that closes the phase covers the LDS writes)
            const int k0 = kbeg + t * HK;
#pragma unroll
            for (int i = 0; i < TA::NI; ++i)
                *reinterpret_cast<uint4*>(As + (wave * TA::NI + i) * 1024 + 16 * lane) = TA::tail(g.A, g.lda, g.mnA, m0, k0, wave * TA::NI + i, lane, g.kvalidA);
#pragma unroll
            for (int i = 0; i < TB::NI; ++i)
                *reinterpret_cast<uint4*>(Bs + (wave * TB::NI + i) * 1024 + 16 * lane) = TB::tail(g.B, g.ldb, g.mnB, n0, k0, wave * TB::NI + i, lane, g.kvalidB);
        }
    };
    // wait until at most `tiles` whole tiles' worth of this wave's DMA pieces are outstanding (+ every LDS access of the wave retired)
    auto wait_tiles = [&](int tiles) {
        if (tiles >= 3) h_wait<3 * NP>();
        else if (tiles == 2) h_wait<2 * NP>();
        else if (tiles == 1) h_wait<NP>();
        else h_wait<0>();
    };

    const int grp = wave >> 2;
    bf16x8 fra[2][TM], frb[2][TN];

    const int npre = min(nk, HAHEAD);
    for (int t = 0; t < npre; ++t) issue(t);
    wait_tiles(npre - 1);                                          // tile 0 landed
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                  // the stagger: group 1 one slot behind from here on
    int slot_out = 0;
    for (int t = 0; t < nk; ++t) {
        const char* As = ringA + slot_out * SA;
        const char* Bs = ringB + slot_out * SB;
        slot_out = slot_out + 1 == HSLOTS ? 0 : slot_out + 1;
        // ---- L(t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fra[u][i] = fa.get(As, i, half, u);
#pragma unroll
            for (int j = 0; j < TN; ++j) frb[u][j] = fb.get(Bs, j, half, u);
        }
        if (t + HAHEAD < nk) issue(t + HAHEAD);
        wait_tiles(min(nk - 2 - t, HAHEAD - 1));                   // tile t+1 landed (tiles t+2 .. t+4, as far as they exist, may be in flight)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- M(t)
        if constexpr (!A_KC) {
            if (do_asum) {          // wave-uniform: first tile column, first wave column (row sums of A: the bias gradient)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float s = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) s += (float)fra[u][i][e];
                        asum[i] += s;
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
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
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
    bf_epilogue<TM, TN, CT>(g, acc, m0 + wm0, n0 + wn0, l31, half, tc.split);
}

template <typename CT>
void bf_launch_h(const BfArgs& g, int big, int transA, int transB, hipStream_t s) {
    const dim3 grid((unsigned)(g.ntiles * g.splits)), blk(512);
#define YT_BFH(BMV, WPSV)                                                                                                          \
    do {                                                                                                                           \
        if (!transA && transB) hipLaunchKernelGGL((gemm_bf16_h_kernel<BMV, BMV, true, true, 8, WPSV, CT>), grid, blk, 0, s, g);          \
        else if (!transA && !transB) hipLaunchKernelGGL((gemm_bf16_h_kernel<BMV, BMV, true, false, 8, WPSV, CT>), grid, blk, 0, s, g);   \
        else if (transA && !transB) hipLaunchKernelGGL((gemm_bf16_h_kernel<BMV, BMV, false, false, 8, WPSV, CT>), grid, blk, 0, s, g);   \
        else hipLaunchKernelGGL((gemm_bf16_h_kernel<BMV, BMV, false, true, 8, WPSV, CT>), grid, blk, 0, s, g);                           \
    } while (0)
    if (big) YT_BFH(256, 2);
    else YT_BFH(128, 4);
#undef YT_BFH
}
template void bf_launch_h<float>(const BfArgs&, int, int, int, hipStream_t);
template void bf_launch_h<bf16_t>(const BfArgs&, int, int, int, hipStream_t);

}  // namespace ytvln
