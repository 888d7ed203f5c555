// Loss kernels of the pre-training / fine-tuning loops (utils/utils_init.py:108-164): ignore-index cross entropy over the
// 30522-way language logits (and the K-way ranking logits with -inf padding), the masked KL of the 1601-way vision head and
// the pos-weighted BCE of the trajectory head.  One workgroup per row streams the row once (online log-sum-exp); the
// scalar reduction is a fixed-order single-workgroup pass, so results are run-to-run deterministic and need no host sync
// (the reference calls .item() here, utils_init.py:127).
#include "common.h"
#include <algorithm>

namespace ytvln {

struct MS { float m, s; };
__device__ __forceinline__ MS ms_combine(MS a, MS b) {
    const float m = fmaxf(a.m, b.m);
    if (m == -INFINITY) return {m, 0.f};
    return {m, a.s * expf(a.m - m) + b.s * expf(b.m - m)};
}
__device__ __forceinline__ MS ms_push(MS a, float x) {
    if (x == -INFINITY) return a;
    if (x <= a.m) return {a.m, a.s + expf(x - a.m)};
    return {x, a.s * expf(a.m - x) + 1.0f};
}
// logits are fp32, or bf16 on the bf16-resident path (BASELINE configs[4]: the decoders write bf16 logits, the gradient goes back as bf16)
__device__ __forceinline__ float lget(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float lget(const uint16_t* p, int64_t i) { return __uint_as_float((uint32_t)p[i] << 16); }

// block-wide (256 threads) log-sum-exp of a row; result valid in all threads
template <typename LT>
__device__ __forceinline__ float block_lse(const LT* __restrict__ row, int V, float* sh /* [8] */) {
    MS a = {-INFINITY, 0.f};
    for (int c = threadIdx.x; c < V; c += 256) a = ms_push(a, lget(row, c));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MS b = {__shfl_xor(a.m, o, 64), __shfl_xor(a.s, o, 64)};
        a = ms_combine(a, b);
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[2 * w] = a.m; sh[2 * w + 1] = a.s; }
    __syncthreads();
    MS r = {sh[0], sh[1]};
#pragma unroll
    for (int i = 1; i < 4; ++i) r = ms_combine(r, MS{sh[2 * i], sh[2 * i + 1]});
    return r.m + logf(r.s);
}
__device__ __forceinline__ float block_sum(float v, float* sh /* [4] */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename LT>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const LT* __restrict__ logits, int64_t ld, const int64_t* __restrict__ target,
                                                     int64_t ignore, float* __restrict__ row_lse, float* __restrict__ row_loss, int V) {
    __shared__ float sh[8];
    const int row = blockIdx.x;
    const int64_t t = target[row];
    if (t == ignore) {          // ~85 % of the masked-LM rows: their loss is 0 and ce_bwd_kernel never reads their log-sum-exp (block-uniform exit)
        if (threadIdx.x == 0) { row_lse[row] = 0.f; row_loss[row] = 0.f; }
        return;
    }
    const LT* x = logits + (int64_t)row * ld;
    const float lse = block_lse(x, V, sh);
    if (threadIdx.x == 0) {
        row_lse[row] = lse;
        // a target outside [0, V) that is not the ignore index (torch's kernel asserts): never read out of bounds, poison the loss
        row_loss[row] = (t == ignore) ? 0.f : (t >= 0 && t < V) ? lse - lget(x, t) : NAN;
    }
}

// out[0] = sum(row_loss) / count, out[1] = count, count = #(target != ignore)  (0/0 = NaN like F.cross_entropy)
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ target,
                                                          int64_t ignore, int M, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f, c = 0.f;
    for (int r = threadIdx.x; r < M; r += 256) { s += row_loss[r]; c += (target[r] != ignore) ? 1.f : 0.f; }
    const float S = block_sum(s, sh);
    const float C = block_sum(c, sh);
    if (threadIdx.x == 0) { out[0] = S / C; out[1] = C; }
}

// DT = float: fp32 logits, fp32 gradient (columns [0, V)).  DT = uint16_t: the bf16-resident path -- bf16 logits, the gradient rounded to bf16
// (RNE) AND zeros in the padding columns [V, ldd), so that the buffer feeds the bf16 GEMMs as a zero-padded operand (YTVLN_GEMM_A_ZERO_PADDED).
__device__ __forceinline__ void grad_store(float* d, int c, float v) { d[c] = v; }
__device__ __forceinline__ void grad_store(uint16_t* d, int c, float v) { d[c] = __builtin_bit_cast(uint16_t, (__bf16)v); }

template <typename DT>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const DT* __restrict__ logits, int64_t ld, const int64_t* __restrict__ target,
                                                     int64_t ignore, const float* __restrict__ row_lse, const float* __restrict__ out,
                                                     const float* __restrict__ gout, DT* __restrict__ dl, int64_t ldd, int V) {
    const int row = blockIdx.x;
    const int64_t t = target[row];
    DT* d = dl + (int64_t)row * ldd;
    const int Vz = sizeof(DT) == 2 ? (int)ldd : V;          // bf16: the padding is zeroed too
    if (t == ignore) {
        for (int c = threadIdx.x; c < Vz; c += 256) grad_store(d, c, 0.f);
        return;
    }
    const float coef = gout[0] / out[1], lse = row_lse[row];
    const DT* x = logits + (int64_t)row * ld;
    for (int c = threadIdx.x; c < V; c += 256) grad_store(d, c, (expf(lget(x, c) - lse) - (c == t ? 1.f : 0.f)) * coef);
    for (int c = V + threadIdx.x; c < Vz; c += 256) grad_store(d, c, 0.f);
}

template <typename LT>
__global__ __launch_bounds__(256) void kl_fwd_kernel(const LT* __restrict__ pred, int64_t ld, const float* __restrict__ tgt, int64_t ldt,
                                                     const int64_t* __restrict__ mask, float* __restrict__ row_lse,
                                                     float* __restrict__ row_loss, int C) {
    __shared__ float sh[8];
    const int row = blockIdx.x;
    if (mask[row] == 0) {
        if (threadIdx.x == 0) { row_lse[row] = 0.f; row_loss[row] = 0.f; }
        return;
    }
    const LT* x = pred + (int64_t)row * ld;
    const float* t = tgt + (int64_t)row * ldt;
    const float lse = block_lse(x, C, sh);
    float acc = 0.f;   // sum_c t * (log t - (x - lse)), 0 where t == 0 (xlogy semantics of F.kl_div)
    for (int c = threadIdx.x; c < C; c += 256) {
        const float tv = t[c];
        if (tv > 0.f) acc += tv * (logf(tv) - (lget(x, c) - lse));
    }
    const float tot = block_sum(acc, sh) * (float)mask[row];
    if (threadIdx.x == 0) { row_lse[row] = lse; row_loss[row] = tot; }
}

// out[0] = sum(row_loss) / max(1, sum(mask)), out[1] = max(1, sum(mask))   (utils_init.py:126-128)
__global__ __launch_bounds__(256) void kl_finalize_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ mask, int M,
                                                          float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f, c = 0.f;
    for (int r = threadIdx.x; r < M; r += 256) { s += row_loss[r]; c += (float)mask[r]; }
    const float S = block_sum(s, sh);
    const float Cn = fmaxf(1.f, block_sum(c, sh));
    if (threadIdx.x == 0) { out[0] = S / Cn; out[1] = Cn; }
}

template <typename DT>
__global__ __launch_bounds__(256) void kl_bwd_kernel(const DT* __restrict__ pred, int64_t ld, const float* __restrict__ tgt, int64_t ldt,
                                                     const int64_t* __restrict__ mask, const float* __restrict__ row_lse,
                                                     const float* __restrict__ out, const float* __restrict__ gout,
                                                     DT* __restrict__ dp, int64_t ldd, int C) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    DT* d = dp + (int64_t)row * ldd;
    const int Cz = sizeof(DT) == 2 ? (int)ldd : C;
    if (mask[row] == 0) {
        for (int c = threadIdx.x; c < Cz; c += 256) grad_store(d, c, 0.f);
        return;
    }
    const DT* x = pred + (int64_t)row * ld;
    const float* t = tgt + (int64_t)row * ldt;
    float ts = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) ts += t[c];
    const float tsum = block_sum(ts, sh);
    const float coef = gout[0] / out[1] * (float)mask[row], lse = row_lse[row];
    for (int c = threadIdx.x; c < C; c += 256) grad_store(d, c, coef * (expf(lget(x, c) - lse) * tsum - t[c]));
    for (int c = C + threadIdx.x; c < Cz; c += 256) grad_store(d, c, 0.f);
}

__device__ __forceinline__ float softplus_neg(float x) { return log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f); }   // log(1+exp(-x))

__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ pw,
                                                      float* __restrict__ out, int n) {
    __shared__ float sh[4];
    const float w = pw ? pw[0] : 1.f;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float lw = 1.f + (w - 1.f) * t[i];
        acc += (1.f - t[i]) * x[i] + lw * softplus_neg(x[i]);
    }
    const float S = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = S / (float)n;
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ pw,
                                                      const float* __restrict__ gout, float* __restrict__ dx, int n) {
    const float w = pw ? pw[0] : 1.f, coef = gout[0] / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float lw = 1.f + (w - 1.f) * t[i];
        const float sg = 1.f / (1.f + expf(x[i]));   // sigmoid(-x)
        dx[i] = ((1.f - t[i]) - lw * sg) * coef;
    }
}

}  // namespace ytvln

using namespace ytvln;

extern "C" int ytvln_ce_fwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index, float* row_lse,
                                float* row_loss, float* out, int M, int V, void* stream) {
    YT_REQUIRE(logits && target && row_lse && row_loss && out && M > 0 && V > 0 && ld >= V, "ce_fwd: bad argument");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(M), dim3(256), 0, s, logits, ld, target, ignore_index, row_lse, row_loss, V);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, row_loss, target, ignore_index, M, out);
    YT_LAUNCH_CHECK("ce_fwd");
    return 0;
}

extern "C" int ytvln_ce_fwd_bf16(const uint16_t* logits, int64_t ld, const int64_t* target, int64_t ignore_index, float* row_lse,
                                 float* row_loss, float* out, int M, int V, void* stream) {
    YT_REQUIRE(logits && target && row_lse && row_loss && out && M > 0 && V > 0 && ld >= V, "ce_fwd_bf16: bad argument");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(ce_fwd_kernel<uint16_t>, dim3(M), dim3(256), 0, s, logits, ld, target, ignore_index, row_lse, row_loss, V);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, row_loss, target, ignore_index, M, out);
    YT_LAUNCH_CHECK("ce_fwd_bf16");
    return 0;
}

extern "C" int ytvln_ce_bwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index, const float* row_lse,
                                const float* out, const float* gout, float* dlogits, int64_t ldd, int M, int V, void* stream) {
    YT_REQUIRE(logits && target && row_lse && out && gout && dlogits && M > 0 && V > 0 && ldd >= V, "ce_bwd: bad argument");
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(M), dim3(256), 0, as_stream(stream), logits, ld, target, ignore_index, row_lse, out, gout,
                       dlogits, ldd, V);
    YT_LAUNCH_CHECK("ce_bwd");
    return 0;
}

extern "C" int ytvln_ce_bwd_bf16(const uint16_t* logits, int64_t ld, const int64_t* target, int64_t ignore_index, const float* row_lse,
                                 const float* out, const float* gout, uint16_t* dlogits, int64_t ldd, int M, int V, void* stream) {
    YT_REQUIRE(logits && target && row_lse && out && gout && dlogits && M > 0 && V > 0 && ldd >= V, "ce_bwd_bf16: bad argument");
    hipLaunchKernelGGL(ce_bwd_kernel<uint16_t>, dim3(M), dim3(256), 0, as_stream(stream), logits, ld, target, ignore_index, row_lse, out, gout,
                       dlogits, ldd, V);
    YT_LAUNCH_CHECK("ce_bwd_bf16");
    return 0;
}

extern "C" int ytvln_kl_fwd_f32(const float* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask, float* row_lse,
                                float* row_loss, float* out, int M, int C, void* stream) {
    YT_REQUIRE(pred && target && mask && row_lse && row_loss && out && M > 0 && C > 0, "kl_fwd: bad argument");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(kl_fwd_kernel<float>, dim3(M), dim3(256), 0, s, pred, ld, target, ldt, mask, row_lse, row_loss, C);
    hipLaunchKernelGGL(kl_finalize_kernel, dim3(1), dim3(256), 0, s, row_loss, mask, M, out);
    YT_LAUNCH_CHECK("kl_fwd");
    return 0;
}

extern "C" int ytvln_kl_fwd_bf16(const uint16_t* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask, float* row_lse,
                                 float* row_loss, float* out, int M, int C, void* stream) {
    YT_REQUIRE(pred && target && mask && row_lse && row_loss && out && M > 0 && C > 0, "kl_fwd_bf16: bad argument");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(kl_fwd_kernel<uint16_t>, dim3(M), dim3(256), 0, s, pred, ld, target, ldt, mask, row_lse, row_loss, C);
    hipLaunchKernelGGL(kl_finalize_kernel, dim3(1), dim3(256), 0, s, row_loss, mask, M, out);
    YT_LAUNCH_CHECK("kl_fwd_bf16");
    return 0;
}

extern "C" int ytvln_kl_bwd_f32(const float* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask,
                                const float* row_lse, const float* out, const float* gout, float* dpred, int64_t ldd, int M, int C,
                                void* stream) {
    YT_REQUIRE(pred && target && mask && row_lse && out && gout && dpred && M > 0 && C > 0, "kl_bwd: bad argument");
    hipLaunchKernelGGL(kl_bwd_kernel<float>, dim3(M), dim3(256), 0, as_stream(stream), pred, ld, target, ldt, mask, row_lse, out, gout, dpred,
                       ldd, C);
    YT_LAUNCH_CHECK("kl_bwd");
    return 0;
}

extern "C" int ytvln_kl_bwd_bf16(const uint16_t* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask, const float* row_lse,
                                 const float* out, const float* gout, uint16_t* dpred, int64_t ldd, int M, int C, void* stream) {
    YT_REQUIRE(pred && target && mask && row_lse && out && gout && dpred && M > 0 && C > 0 && ldd >= C, "kl_bwd_bf16: bad argument");
    hipLaunchKernelGGL(kl_bwd_kernel<uint16_t>, dim3(M), dim3(256), 0, as_stream(stream), pred, ld, target, ldt, mask, row_lse, out, gout, dpred,
                       ldd, C);
    YT_LAUNCH_CHECK("kl_bwd_bf16");
    return 0;
}

extern "C" int ytvln_bce_fwd_f32(const float* x, const float* t, const float* pos_weight, float* out, int n, void* stream) {
    YT_REQUIRE(x && t && out && n > 0 && n <= 65536, "bce_fwd: bad argument (n=%d)", n);
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, t, pos_weight, out, n);
    YT_LAUNCH_CHECK("bce_fwd");
    return 0;
}

extern "C" int ytvln_bce_bwd_f32(const float* x, const float* t, const float* pos_weight, const float* gout, float* dx, int n,
                                 void* stream) {
    YT_REQUIRE(x && t && gout && dx && n > 0 && n <= 65536, "bce_bwd: bad argument (n=%d)", n);
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, t, pos_weight, gout, dx, n);
    YT_LAUNCH_CHECK("bce_bwd");
    return 0;
}
