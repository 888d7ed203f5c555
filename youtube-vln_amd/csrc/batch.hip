// On-device batch preparation: the BERT / ViLBERT masking of instruction tokens and region features
// that the reference applies per dataset item on the host (utils/dataset/common.py:213-300), as two HBM-bound kernels over
// the whole [bs*K, ...] batch.  The uniform draws come either from the caller (explicit `p` / `random_tokens`: used to pin the
// kernels bit-for-bit against the reference's functions) or from the library's Philox stream (production).
#include "common.h"
#include <algorithm>

namespace ytvln {

// float thresholds exactly as torch compares an fp32 tensor with the reference's Python doubles (scalar -> fp32)
__device__ __forceinline__ float thr_mask() { return (float)0.85; }
__device__ __forceinline__ float thr_tok_random() { return (float)(0.85 + 0.15 * 0.8); }
__device__ __forceinline__ float thr_tok_keep() { return (float)(0.85 + 0.15 * 0.9); }
__device__ __forceinline__ float thr_reg_zero() { return (float)(0.85 + 0.15 * 0.1); }

// uniform in [0, 1) with 24 random bits (torch.rand's fp32 construction)
__device__ __forceinline__ float u01(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

// randomize_tokens (common.py:213-270 with mask_action_rate == 0):  p = U * mask;  p >= 0.85 -> target = token, token = [MASK];
// p >= 0.97 -> token = random id;  p >= 0.985 -> token = original.  targets = -1 elsewhere.
__global__ __launch_bounds__(256) void randomize_tokens_kernel(const int64_t* __restrict__ tokens, const int64_t* __restrict__ mask,
                                                               int64_t n, int vocab, int64_t mask_id, const float* __restrict__ p_in,
                                                               const int64_t* __restrict__ rand_in, const int64_t* __restrict__ rng,
                                                               int64_t site, int64_t* __restrict__ out, int64_t* __restrict__ targets) {
    DropKey key = {0, 0, 0, 0};
    if (!p_in || !rand_in) key = make_drop_key(rng, site);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t tok = tokens[i];
        u32x4 b = {0, 0, 0, 0};
        if (!p_in || !rand_in) b = drop_bits(key, (uint64_t)i);
        const float p = (p_in ? p_in[i] : u01(b.x)) * (float)mask[i];
        const int64_t rnd = rand_in ? rand_in[i] : (int64_t)(b.y % (uint32_t)vocab);
        int64_t t = tok, tg = -1;
        if (p >= thr_mask()) { tg = tok; t = mask_id; }
        if (p >= thr_tok_random()) t = rnd;
        if (p >= thr_tok_keep()) t = tok;
        out[i] = t;
        targets[i] = tg;
    }
}

// randomize_regions (common.py:272-300): one wave per region row.  p = U * mask;  p >= 0.85 -> targets = probs, targets_mask = 1
// (else targets = 1/C, 0);  p >= 0.865 -> the 2048-d feature row is zeroed in place.
__global__ __launch_bounds__(256) void randomize_regions_kernel(float* __restrict__ feat, int64_t ldf, const float* __restrict__ probs,
                                                                const int64_t* __restrict__ mask, int64_t rows, int F, int C,
                                                                const float* __restrict__ p_in, const int64_t* __restrict__ rng,
                                                                int64_t site, float* __restrict__ targets, int64_t* __restrict__ tmask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    DropKey key = {0, 0, 0, 0};
    if (!p_in) key = make_drop_key(rng, site);
    const float uni = 1.0f / (float)C;            // torch.ones_like(probs) / C in fp32
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const float p = (p_in ? p_in[r] : u01(drop_bits(key, (uint64_t)r).x)) * (float)mask[r];
        const bool sel = p >= thr_mask();
        float* trow = targets + r * (int64_t)C;
        const float* prow = probs + r * (int64_t)C;
        for (int c = lane; c < C; c += 64) trow[c] = sel ? prow[c] : uni;
        if (lane == 0) tmask[r] = sel ? 1 : 0;
        if (p >= thr_reg_zero()) {
            float* frow = feat + r * ldf;
            if ((F & 3) == 0 && (ldf & 3) == 0 && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0)) {
                for (int c = lane; c < (F >> 2); c += 64) reinterpret_cast<float4*>(frow)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int c = lane; c < F; c += 64) frow[c] = 0.f;
            }
        }
    }
}

}  // namespace ytvln

using namespace ytvln;

extern "C" int ytvln_randomize_tokens(const int64_t* tokens, const int64_t* mask, int64_t n, int vocab_size, int64_t mask_token_id,
                                      const float* p, const int64_t* random_tokens, const int64_t* rng, int64_t site,
                                      int64_t* tokens_out, int64_t* targets_out, void* stream) {
    YT_REQUIRE(tokens && mask && tokens_out && targets_out && n >= 0 && vocab_size > 0, "randomize_tokens: bad argument");
    YT_REQUIRE((p && random_tokens) || rng, "randomize_tokens: needs explicit draws (p, random_tokens) or the rng state");
    if (n == 0) return 0;
    hipLaunchKernelGGL(randomize_tokens_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0, as_stream(stream), tokens, mask,
                       n, vocab_size, mask_token_id, p, random_tokens, rng, site, tokens_out, targets_out);
    YT_LAUNCH_CHECK("randomize_tokens");
    return 0;
}

extern "C" int ytvln_randomize_regions(float* features, int64_t ldf, const float* probs, const int64_t* mask, int64_t rows, int F, int C,
                                       const float* p, const int64_t* rng, int64_t site, float* targets, int64_t* targets_mask,
                                       void* stream) {
    YT_REQUIRE(features && probs && mask && targets && targets_mask && rows >= 0 && F > 0 && C > 0 && ldf >= F, "randomize_regions: bad argument");
    YT_REQUIRE(p || rng, "randomize_regions: needs explicit draws (p) or the rng state");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(randomize_regions_kernel, dim3((unsigned)std::min<int64_t>(cdiv(rows, 4), 8192)), dim3(256), 0, as_stream(stream), features,
                       ldf, probs, mask, rows, F, C, p, rng, site, targets, targets_mask);
    YT_LAUNCH_CHECK("randomize_regions");
    return 0;
}
