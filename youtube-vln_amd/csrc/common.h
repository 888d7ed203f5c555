// Shared host/device helpers for libytvln (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include <type_traits>
#include <utility>

#include "../../include/ytvln.h"

namespace ytvln {

// ---- error reporting across the C ABI -------------------------------------------------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);

#define YT_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return ::ytvln::fail(-1, __VA_ARGS__);    \
    } while (0)

#define YT_LAUNCH_CHECK(name)                                                                   \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) return ::ytvln::fail(-2, "%s: %s", name, hipGetErrorString(e_));  \
    } while (0)

// ---- run-time options (ytvln_set_option / ytvln_get_option; misc.hip holds the table) ----------------------------------------------
// The complete list of switches the library reads.  Each one starts from the environment variable YTVLN_<NAME> (read once, at first
// use) and can be changed at run time through the C ABI; INTEGRATION.md documents every entry and tests/test_abi.py checks that its
// table and this one agree.
enum Option {
    OPT_ATTN_W1 = 0,         // bit mask of the one-wave-per-SIMD attention kernels: 1 forward, 2 dQ, 4 dK/dV (default 7; 0 = two-wave forms)
    OPT_ATTN_W1_DKV_ANY,     // 1: the one-wave dK/dV kernel for launches of any size (default 0: only when its last round of wave slots is >= 85 % full)
    OPT_ATTN_DSPLIT,         // 1 (default): two-wave forward workgroups holding a single query tile split the head dimension between their waves
    OPT_GEMM_TILE,           // -1 (default): planner; 0..4 forces 128x128 / 128x64 / 64x64 / 256x128 / 256x256 where legal
    OPT_GEMM_SPLITS,         // -1 (default): planner; n forces the split-K count where split-K is legal
    OPT_GEMM_SPLIT_MAP,      // 1 (default): split-K workgroups laid out split-major per XCD; 0: split-fastest (same results, more fabric traffic)
    OPT_GEMM_GENERIC,        // 1: every fp32 GEMM on the register-staged generic kernel (default 0)
    OPT_GEMM_SW,             // one-wave-per-SIMD fp32 GEMM kernels (gemm_sw.hip) for the 256-row tiles: 0 never, 1 every such launch, 2 only launches without split-K, 3 the 256x256 launches whose B operand is [K, N]
    OPT_GEMM_SK,             // persistent fp32 GEMM (gemm_sk.hip): 0 (default) never -- it measured 3-10 % behind the launch-per-tile kernel on every cfg-2 shape
                             // (profiles/round5_gemm_forms.log) --, 1 where its cost model wins, 2 whole-tile (DP) form wherever legal, 3 stream-K form wherever legal
    OPT_GEMM_SK_TILE,        // -1 (default): planner; 3 / 4 forces 256x128 / 256x256 tiles for the persistent kernel
    OPT_GEMM_SK_GROUPS,      // 8 (default): one ticket group per XCD; 1: one group over the whole launch (partial tiles may cross XCDs)
    OPT_GEMM_T224,           // 1 (default): the fp32 planner may take the 224x256 tile (eight waves of 224x32) for single-round launches with a K-contiguous A; 0: never
    OPT_GEMM_BF16_FORM,      // bf16 GEMM main loop: -1 (default) chosen per launch, 0 eight waves / DMA in the load phases, 1-2 DMA between the matrix instructions, 3 32-deep tiles, 4 four waves of 128x128
    OPT_GEMM_STAGGER,        // bf16 GEMM, launches of >= 3 rounds of 256x256 tiles: half of the first round starts this many percent of a tile's main loop late (0 = off)
    OPT_ATTN_DKV_SPLIT,      // bf16 attention, d = 128: 1 = dK and dV in two passes with two waves per SIMD each, 0 = one pass with one wave per SIMD
    OPT_GEMM_BF16_WIDE,      // 1: bf16 GEMM outputs leave through LDS-transposed 16-byte stores where the epilogue reads no matrix; 0: 2-byte stores
    OPT_COUNT
};
int opt(int id);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}).  Register arrays indexed through it
// stay in registers (a run-time index, even of a fully unrolled loop variable captured by a lambda, can demote them to scratch).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- Philox4x32-10 counter RNG (dropout masks reproducible between forward and backward) ----------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return {c0, c1, c2, c3};
}

// Key of one dropout site: rng[0] = seed, rng[1] = forward counter (device memory), site = host call-site id.
struct DropKey { uint32_t k0, k1, s0, s1; };
__device__ __forceinline__ DropKey drop_key_of(uint64_t seed, uint64_t ctr, int64_t site) {      // (seed, ctr) already loaded by the caller
    const uint64_t mix = seed ^ (ctr * 0x9E3779B97F4A7C15ull);
    return {(uint32_t)mix, (uint32_t)(mix >> 32), (uint32_t)site, (uint32_t)((uint64_t)site >> 32) ^ (uint32_t)(ctr >> 17)};
}
__device__ __forceinline__ DropKey make_drop_key(const int64_t* rng, int64_t site) { return drop_key_of((uint64_t)rng[0], (uint64_t)rng[1], site); }
// Four keep-flags/uniforms for elements [4*q, 4*q+3] of the site's flat element space.
__device__ __forceinline__ u32x4 drop_bits(const DropKey& k, uint64_t q) {
    return philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), k.s0, k.s1, k.k0, k.k1);
}
__device__ __forceinline__ uint32_t drop_threshold(float p) {
    // keep iff bits >= thr ; P(drop) = thr / 2^32
    double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
// keep-scale for a single element e of the site (recomputes the 4-group; use the vector form on hot paths)
__device__ __forceinline__ float drop_scale1(const DropKey& k, uint64_t e, uint32_t thr, float inv_keep) {
    const u32x4 b = drop_bits(k, e >> 2);
    const uint32_t sel = (e & 3) == 0 ? b.x : (e & 3) == 1 ? b.y : (e & 3) == 2 ? b.z : b.w;
    return sel >= thr ? inv_keep : 0.0f;
}

// Cheap per-element keep decision for attention-probability dropout: one 32-bit hash per score, the same in every fragment layout (the forward
// / dQ kernels hold 4 consecutive keys per lane, the dK/dV kernel 4 consecutive queries).  The element index enters through two odd multipliers
// (lo * C1 is an add per element for the kernels: lo advances by compile-time steps), the site / step key -- itself put through a full avalanche
// once per wave, so that neighbouring sites and steps get unrelated keys -- by xor; then five FULL-RATE operations: a 24-bit multiply-add
// (v_mad_u32_u24, fed by the low 24 bits, the high 24 added back), an xor-shift, a 24-bit multiply.  Round 3: replaces the murmur3 finaliser
// applied per element (2 quarter-rate v_mul_lo_u32 + 6 plain ops = 56 issue cycles of the ~100 a score element cost each attention kernel;
// now 20) -- a wave's VALU time is not hidden under its matrix instructions (LABNOTES.md 5c).  Quality, measured against the finaliser on
// 4096 x 288 masks at p = 0.1 over 60 random key pairs: drop rate 0.1000 (0.0992-0.1006 per mask), correlations between neighbours along keys /
// rows / diagonals / +32 keys <= 0.003, correlation between the masks of two keys 0.0009 rms / 0.0033 max -- all at the sampling-noise floor
// (0.0009), as for the finaliser.  (A version with ONE 32-bit multiply also passes the single-mask tests but leaves 0.0025 rms / 0.013 max
// between masks of different keys: rejected.)
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t attn_drop_hash(uint32_t lo, uint32_t hi, const DropKey& k) {
    const uint32_t kx = fmix32(k.k0 ^ fmix32((k.s0 + 0x9E3779B9u) ^ k.k1));      // (wave-invariant: computed once)
    uint32_t x = (lo * 0x9E3779B1u) ^ (hi * 0x85EBCA77u) ^ kx;
    x = (uint32_t)__umul24(x, 0x6B43A9u) + (x >> 8);
    x ^= x >> 12;
    return (uint32_t)__umul24(x, 0xB5297Au);
}

// ---- wave / block reductions ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact erf-GELU (vilbert/vilbert.py:119) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// The same two functions for the bf16-RESIDENT path, whose every result is rounded to bf16 (8 significant bits) on its way out: Phi(x) by
// Abramowitz-Stegun 7.1.26 -- |error| < 6e-7 in fp32 arithmetic against the exact function (checked over [-12, 12], four orders under the
// rounding that follows), branch-free: one v_rcp, one v_exp, six fma.  The library's erff is two polynomial branches that a wave with mixed
// arguments runs both of (~50 vector instructions per value): on 129024 x 1024 outputs that was 170 us of a 445 us launch
// (profiles/round6_gemm_bf16_epilogues.log).  The fp32 path keeps erff: it owes the reference 1e-4 on logits, not 4e-3.
__device__ __forceinline__ float phi_as(float x, float& e) {          // Phi(x); e = exp(-x^2 / 2) for the density
    const float az = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    e = __expf(-az * az);
    const float h = 0.5f * (p * t) * e;          // = 0.5 erfc(|x| / sqrt 2): the small side, no cancellation for negative x
    return x < 0.f ? h : 1.0f - h;
}
__device__ __forceinline__ float gelu_erf_b(float x) { float e; return x * phi_as(x, e); }
__device__ __forceinline__ float dgelu_erf_b(float x) { float e; const float cdf = phi_as(x, e); return fmaf(x * 0.39894228040143267794f, e, cdf); }

// ---- memory operations of the interior epilogue, written out: scalar row base + one 32-bit per-lane byte offset -------------------------------
// (hipcc turns the C++ form of "uniform pointer + per-lane offset" into 64-bit per-lane address arithmetic or, through integer casts, into FLAT
// accesses; the saddr form needs no address registers at all.)  Loads issued this way are invisible to the compiler's s_waitcnt insertion:
// epi_wait<N>() is the explicit wait and names the loaded registers so that no use can be scheduled above it.  The leading `s_nop 4`: a scalar
// base that the compiler has just produced with a VALU instruction (v_readfirstlane, or v_readlane when it reloads a spilled SGPR) needs five
// wait states before a memory instruction may read it, and the hazard recogniser does not look inside inline asm (round 5: wild addresses,
// "memory aperture violation", exactly in the epilogues with enough scalar pressure to spill).
// a wave-uniform pointer pinned into scalar registers (callers whose wave index is not provably uniform to the compiler, e.g. tid >> 6)
template <class T>
__device__ __forceinline__ T* scalar_ptr(T* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void epi_store(uint32_t voff, float v, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void epi_load(float& d, uint32_t voff, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
// 2-byte forms for bf16 rows (the bf16-resident GEMM's C and aux): the value sits in the low half of the register
__device__ __forceinline__ void epi_store_short(uint32_t voff, uint32_t v, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_store_short %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void epi_load_ushort(float& d, uint32_t voff, const void* sbase) {      // d receives the zero-extended 16 bits (as raw bits)
    asm volatile("s_nop 4\n\tglobal_load_ushort %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void epi_wait(float (&a)[4], float (&b)[4]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, round-robin dispatch): consecutive remapped ids share an
// XCD (and therefore an L2).  cdna guide T1 (bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace ytvln
