// Fused AdamW over flat parameter / gradient / moment arenas (vilbert/optimization.py:141-187, correct_bias=True).
//
// The reference issues ~8 ATen kernels for each of 541 parameter tensors per step; here one launch streams the four
// arenas once: 28 bytes per parameter (read p,g,m,v; write p,m,v), the HBM floor for this update.  Tensors the reference
// skips (grad is None: never-used heads, vilbert_init/optimization.py:143-144) are simply absent from the chunk table,
// so they receive neither state nor decay.  Hyper-parameters live in device memory so a captured hipGraph can be
// replayed while the host updates the learning rate.
#include "common.h"

namespace ytvln {

struct AdamChunk { int64_t off; int64_t len; float wd; float pad; };
static_assert(sizeof(AdamChunk) == 24, "chunk record layout is part of the ABI");

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float ss, float lrwd) {
    m = m * b1 + (1.0f - b1) * g;              // exp_avg.mul_(beta1).add_(1-beta1, grad)            :166
    v = v * b2 + ((1.0f - b2) * g) * g;        // exp_avg_sq.mul_(beta2).addcmul_(1-beta2, grad, grad) :167
    const float denom = sqrtf(v) + eps;        //                                                      :168
    p = p + (-ss) * (m / denom);               // p.addcdiv_(-step_size, exp_avg, denom)               :176
    if (lrwd != 0.f) p = p + (-lrwd) * p;      // p.add_(-lr*wd, p)  -- decay AFTER the update         :186-187
}

// PB != nullptr: the updated parameter is ALSO written as bf16 (round to nearest even) at the same offset of a bf16 arena -- the weight
// operands of the bf16-resident path (BASELINE configs[4]) are refreshed by the optimizer step itself: +2 bytes per parameter, no cast pass.
__device__ __forceinline__ uint32_t bf16_bits(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f); }

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ Mo,
                                                    float* __restrict__ Vo, const AdamChunk* __restrict__ chunks,
                                                    const float* __restrict__ hyper, float gscale, uint16_t* __restrict__ PB) {
    const AdamChunk c = chunks[blockIdx.x];
    const float b1 = hyper[0], b2 = hyper[1], eps = hyper[2], ss = hyper[3], lr = hyper[4];
    const float lrwd = lr * c.wd;
    float* p = P + c.off; const float* g = G + c.off; float* m = Mo + c.off; float* v = Vo + c.off;
    const int64_t n4 = ((c.off & 3) == 0) ? (c.len >> 2) : 0;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        adam1(pv.x, gv.x * gscale, mv.x, vv.x, b1, b2, eps, ss, lrwd);
        adam1(pv.y, gv.y * gscale, mv.y, vv.y, b1, b2, eps, ss, lrwd);
        adam1(pv.z, gv.z * gscale, mv.z, vv.z, b1, b2, eps, ss, lrwd);
        adam1(pv.w, gv.w * gscale, mv.w, vv.w, b1, b2, eps, ss, lrwd);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
        if (PB) reinterpret_cast<uint2*>(PB + c.off)[i] = make_uint2(bf16_bits(pv.x) | (bf16_bits(pv.y) << 16), bf16_bits(pv.z) | (bf16_bits(pv.w) << 16));
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < c.len; i += 256) {
        adam1(p[i], g[i] * gscale, m[i], v[i], b1, b2, eps, ss, lrwd);
        if (PB) PB[c.off + i] = (uint16_t)bf16_bits(p[i]);
    }
}

}  // namespace ytvln

using namespace ytvln;

extern "C" int ytvln_adamw_f32(float* p, const float* g, float* m, float* v, const void* chunks, int nchunks, const float* hyper,
                               float grad_scale, void* stream) {
    YT_REQUIRE(p && g && m && v && chunks && hyper, "adamw: null pointer");
    YT_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw: arenas must be 16-byte aligned");
    if (nchunks <= 0) return 0;
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, as_stream(stream), p, g, m, v,
                       reinterpret_cast<const AdamChunk*>(chunks), hyper, grad_scale, (uint16_t*)nullptr);
    YT_LAUNCH_CHECK("adamw");
    return 0;
}

extern "C" int ytvln_adamw_f32_bf16copy(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, const void* chunks, int nchunks,
                                        const float* hyper, float grad_scale, void* stream) {
    YT_REQUIRE(p && g && m && v && p_bf16 && chunks && hyper, "adamw_bf16copy: null pointer");
    YT_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)p_bf16) & 15) == 0, "adamw_bf16copy: arenas must be 16-byte aligned");
    if (nchunks <= 0) return 0;
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, as_stream(stream), p, g, m, v,
                       reinterpret_cast<const AdamChunk*>(chunks), hyper, grad_scale, p_bf16);
    YT_LAUNCH_CHECK("adamw_bf16copy");
    return 0;
}
