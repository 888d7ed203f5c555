/*
 * ytvln.h -- C ABI of libytvln.so: the MI355X (gfx950 / CDNA4) kernels behind the ViLBERT hot path of
 * JeremyLinky/YouTube-VLN (forward/backward of vilbert/vilbert.py, the losses of utils/utils_init.py:108-164 and
 * the AdamW step of vilbert/optimization.py:141-187).
 *
 * The reference has no FFI layer of its own: its "plugin API" for this path is the Python class surface of
 * vilbert.vilbert / lily.  Each entry point below therefore cites the reference code whose
 * arithmetic it replaces; the Python modules in youtube-vln_amd/ytvln keep the reference class / argument names and
 * call these through ctypes (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (PyTorch allocator);
 *     the library never allocates, frees or synchronises;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - row-major fp32 unless stated; `ld*` are leading dimensions in ELEMENTS;
 *   - return value 0 = launched, negative = rejected (bad argument / launch failure); ytvln_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI;
 *   - dropout: Philox4x32-10 keyed by (rng[0] = seed, rng[1] = forward counter) read from DEVICE memory (graph-replay
 *     safe) and by a host-side `site` id; the backward pass regenerates the identical mask from the same triple.
 */
#ifndef YTVLN_H
#define YTVLN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): ytvln_attn_problem carries `keep` (added in round 5 without a bump: a version-1 binding paired with that library passed the
 * check with a struct 8 bytes short), ytvln_attn_problem_size, ytvln_set_host_wait, ytvln_rccl_allreduce_slices (any dtype). */
#define YTVLN_ABI_VERSION 2

int ytvln_version(void);
const char* ytvln_last_error(void);

/* How the HOST waits for the device (hipSetDeviceFlags on `device`; one process per GPU as in utils/distributed.py:63-104, where N ranks share one
 * host): blocking = 1 -> hipDeviceScheduleBlockingSync, a thread that waits in hipStreamSynchronize / hipDeviceSynchronize / hipEventSynchronize
 * sleeps on an interrupt instead of spinning on a core; 0 -> the runtime's default (spin).  Host-side only; no kernel is affected.  Call it
 * before the first stream / event of the process is created (ytvln.misc.set_host_wait does). */
int ytvln_set_host_wait(int device, int blocking);

/* Run-time options: the complete set of switches the library reads (kernel-form selection for tests and experiments; nothing a
 * production run has to touch).  An option starts from the environment variable YTVLN_<NAME> (read at first use) and can be set at any
 * time; INTEGRATION.md section 4 lists every name, default and meaning.  Names may be given with or without the "YTVLN_" prefix.
 *   ytvln_option_count / ytvln_option_name enumerate the table (index 0 .. count-1); set / get return 0, or -1 for an unknown name. */
int ytvln_option_count(void);
const char* ytvln_option_name(int index);
int ytvln_set_option(const char* name, int value);
int ytvln_get_option(const char* name, int* value);

/* activation / epilogue selectors for ytvln_gemm_f32 */
enum {
    YTVLN_EPI_NONE = 0,       /* C = A.B (+bias)                                                             */
    YTVLN_EPI_GELU = 1,       /* aux = A.B + bias (pre-activation, optional) ; C = gelu_erf(aux)  vilbert.py:119 */
    YTVLN_EPI_RELU = 2,       /* C = max(A.B + bias, 0)                                            vilbert.py:832 */
    YTVLN_EPI_MUL_DGELU = 3,  /* C = (A.B) * gelu'(aux)      backward of EPI_GELU through the next Linear        */
    YTVLN_EPI_MUL_DRELU = 4   /* C = (A.B) * (aux > 0)       backward of EPI_RELU (aux = the forward output)     */
};

/* Dense projection on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).  Replaces every nn.Linear / F.linear on the path
 * (vilbert.py:285-287, 322, 352, 365, 555-568, 641-644, 831, 864, 906, 968, 1358 ...) and their autograd backward.
 *   C[M,N] (+)= op(A)[M,K] . op(B)[K,N]
 *   transA = 0: A stored [M,K] (lda >= K)      transA = 1: A stored [K,M] (lda >= M)
 *   transB = 0: B stored [K,N] (ldb >= N)      transB = 1: B stored [N,K] (ldb >= K)   <- nn.Linear weight layout
 *   bias: [N] or NULL.  beta: 0 = overwrite, 1 = accumulate into C.  aux/ldaux: see the epilogue enum (may be NULL).
 *   workspace: optional scratch of ytvln_gemm_workspace_elems(M,N,K,epilogue) floats.  When it is supplied and the output
 *   has too few 128x128 tiles to fill 256 CUs (weight gradients: [out,in] outputs contracted over N*T or N*R rows), the
 *   contraction is split across workgroups and reduced in a fixed order (deterministic split-K); otherwise ignored.
 *   flags: YTVLN_GEMM_A_ZERO_PADDED = the caller guarantees readable ZERO padding behind A's contiguous dimension up to
 *   lda (rounded to 32 for a K-contiguous A with B = [K,N]; to 4 for an M-contiguous A).  It lets the 30522- and 1601-wide
 *   logit gradients (vilbert.py:906, 968 backward) use the LDS-DMA main loop although 30522 % 32 != 0. */
#define YTVLN_GEMM_A_ZERO_PADDED 1
/* opt-in: every fp32 operand value is split exactly into three bf16 terms in registers and each product is accumulated in fp32
 * from the six largest cross terms on the bf16 matrix instruction (error per product ~ one fp32 rounding; LDS-DMA path only,
 * ignored by the generic kernel).  Finite inputs only: an infinite operand value yields NaN (inf - inf in the
 * split) where the native instruction would propagate the infinity. */
#define YTVLN_GEMM_SPLIT_BF16X3 2
int64_t ytvln_gemm_workspace_elems(int M, int N, int K, int epilogue);
/* Introspection (host only, no GPU work): the tile shape and split count the launch planner picks for an aligned problem of this size
 * (transA = 1: M-contiguous A, which excludes the 256-row tiles).  Used by tests and by tools/ to explain a measurement. */
int ytvln_gemm_plan(int M, int N, int K, int transA, int epilogue, int* tile_m, int* tile_n, int* splits);
/* the same for a launch carrying YTVLN_GEMM_SPLIT_BF16X3 (its planner has its own per-tile costs; 256x256 tiles also for transA = 1 and
 * with split-K) */
int ytvln_gemm_plan_x3(int M, int N, int K, int transA, int epilogue, int* tile_m, int* tile_n, int* splits);
int ytvln_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                   int64_t ldc, const float* bias, float* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                   float beta, float* workspace, int64_t workspace_elems, int flags, void* stream);

/* ytvln_gemm_f32 that ALSO produces a_rowsum[m] = sum_k op(A)[m, k] when it can do so for free: the bias gradient of an nn.Linear
 * (db = sum over rows of dY, vilbert.py:285-287 ... backward) rides on its weight-gradient GEMM dW = dY^T X, whose A operand is dY^T --
 * the workgroups of the first tile column add up the fragments they feed to the matrix cores (fixed order; split-K partials are
 * reduced in split order: deterministic).  *rowsum_done (HOST int) is set to 1 when the sums were produced -- LDS-DMA main loop,
 * M-contiguous fp32 A (transA = 1), K % 32 == 0, no fp32x3 -- and to 0 otherwise (the caller then runs ytvln_colsum_f32).
 * `workspace` as returned by ytvln_gemm_workspace_elems also holds the per-split partial sums. */
int ytvln_gemm_f32_rowsum(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                          int64_t ldc, const float* bias, float* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                          float beta, float* workspace, int64_t workspace_elems, int flags, float* a_rowsum, int* rowsum_done,
                          void* stream);

/* The same projection with the PERSISTENT kernel available (csrc/gemm_sk.hip; the nn.Linear forward and input-gradient GEMMs of
 * vilbert.py:285-287, 322, 352, 365, 555-568, 641-644 with a K-contiguous A): one workgroup per CU walks whole output tiles or --
 * stream-K form -- an equal share of the (tile, k-tile) iteration space, the next piece's operands in flight under the epilogue; partial
 * tiles are added in ascending k order by the workgroup that holds the k = 0 end of the tile (bit-reproducible).
 *   sk_ctl: ytvln_gemm_sk_ctl_elems() uint32 words, ZERO when first used and private to the stream the call is enqueued on (the kernel
 *   leaves it zero); NULL = exactly ytvln_gemm_f32 / ytvln_gemm_f32_rowsum.  workspace (ytvln_gemm_workspace_elems) also holds the
 *   partial tiles.  a_rowsum / rowsum_done: both NULL or both set (as ytvln_gemm_f32_rowsum).  The launch planner decides per shape between
 *   this kernel and the launch-per-tile one (run-time options GEMM_SK, GEMM_SK_TILE, GEMM_SK_GROUPS). */
int64_t ytvln_gemm_sk_ctl_elems(void);
int ytvln_gemm_f32_sk(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                      int64_t ldc, const float* bias, float* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                      float beta, float* workspace, int64_t workspace_elems, int flags, float* a_rowsum, int* rowsum_done,
                      uint32_t* sk_ctl, void* stream);
/* Introspection (host only): does the planner take the persistent kernel for this aligned problem, with which tile, in the whole-tile
 * (*whole_tiles = 1) or the stream-K form, on how many workgroups. */
int ytvln_gemm_sk_plan(int M, int N, int K, int transA, int epilogue, int* use, int* tile_m, int* tile_n, int* whole_tiles,
                       int* workgroups);
/* Diagnostics: when `buffer` is non-NULL every workgroup of a persistent launch writes 16 s_memrealtime stamps (100 MHz) to
 * buffer[16 * workgroup + i]: 0 start, 1 ticket drawn, 2 first operands in LDS, then (main loop done, epilogue issued) per piece, 15 end.
 * The buffer must hold 16 * 1024 uint64 (a launch has at most 2 workgroups per CU); NULL switches the stamps off (default). */
int ytvln_gemm_probe(unsigned long long* buffer);

/* out[b, n] = sum over the b-th block of `rows_per_block` rows of x[:, n].  out is [ceil(M/rows_per_block), N] with
 * leading dimension ldo (bias gradients, position-embedding gradient, second stage of every column reduction). */
int ytvln_colsum_f32(const float* x, int64_t ldx, int M, int N, float* out, int64_t ldo, int rows_per_block,
                     void* stream);

/* out[b, k, n] = sum over rows r of block b with idx[r] == k of x[r, n], for small tables (KT <= 32): gradient of
 * image_sequence_embeddings (vilbert.py:1364) and token_type_embeddings (:251).  idx_f32 (float indices, as stored in
 * image_loc[..., 11]) or idx_i64 is used, whichever is non-NULL. */
int ytvln_colsum_by_index_f32(const float* x, int64_t ldx, const float* idx_f32, int64_t idx_stride,
                              const int64_t* idx_i64, int M, int N, int KT, float* out, int rows_per_block,
                              void* stream);

/* table_grad[idx[r], :] += x[r, :] (atomic), rows with idx == skip_idx ignored: backward of nn.Embedding with
 * padding_idx (vilbert.py:225-227, 249). */
int ytvln_scatter_add_rows_f32(const float* x, int64_t ldx, const int64_t* idx, int M, int H, float* table_grad,
                               int64_t skip_idx, void* stream);
/* Deterministic form for repeated indices (word-embedding gradient, BertEmbeddings backward, vilbert.py:219-256): `sorted_idx` = the
 * indices in ascending order (stable sort), `perm` = the source row of each sorted position.  One wave owns each run of equal indices:
 * no atomics, fixed summation order -> bit-reproducible training steps. */
int ytvln_scatter_add_rows_sorted_f32(const float* x, int64_t ldx, const int64_t* sorted_idx, const int64_t* perm, int M, int H,
                                      float* table_grad, int64_t skip_idx, void* stream);

/* The bf16-resident path (BASELINE configs[4], ytvln.ops.set_matmul_precision("bf16")): activations, gradients and a bf16 copy of the
 * weights live in HBM as bf16 and every projection reads its operands IN PLACE -- no per-call staging, cast or transpose:
 *   C[M,N] (+)= op(A)[M,K] . op(B)[K,N]   with the transA / transB conventions of ytvln_gemm_f32 (bf16 elements, leading dimensions in elements);
 *   an operand whose contraction index is its row (transA = 1, transB = 0) is gathered from LDS by ds_read_b64_tr_b16.
 * fp32 accumulation on v_mfma_f32_32x32x16_bf16; bias is fp32; the epilogue runs in fp32 and rounds once (RNE) into C, which is bf16
 * (c_dtype = YTVLN_DT_BF16) or fp32 (YTVLN_DT_F32: logits, pooled heads, weight gradients); aux (GELU pre-activation out, GELU' / ReLU' in) is
 * bf16.  workspace: ytvln_gemm_bf16_workspace_elems(M,N,K,epilogue) floats (deterministic split-K of the weight gradients); flags:
 * YTVLN_GEMM_A_ZERO_PADDED as for ytvln_gemm_f32, padding in units of 8 elements.  a_rowsum / rowsum_done (may both be NULL): as
 * ytvln_gemm_f32_rowsum -- row sums of a k-major A (the bias gradient riding on the weight-gradient GEMM), fp32.
 * Operands that are not 16-byte aligned with leading dimensions % 8 == 0 (tiny configurations) run a slow generic kernel: same results. */
int64_t ytvln_gemm_bf16_workspace_elems(int M, int N, int K, int epilogue);
int ytvln_gemm_bf16(const uint16_t* A, int64_t lda, int transA, const uint16_t* B, int64_t ldb, int transB, void* C, int64_t ldc,
                    int c_dtype, const float* bias, uint16_t* aux, int64_t ldaux, int M, int N, int K, int epilogue, float beta,
                    float* workspace, int64_t workspace_elems, int flags, float* a_rowsum, int* rowsum_done, void* stream);
/* Diagnostics: when `buffer` (128 uint32, device) is non-NULL, bf16-output forward-layout launches on the 256x256 tile run an instrumented build of
 * the kernel in which waves 0 and 4 of workgroup `block` record s_memtime (shader cycles, low 32 bits) for k-tiles 8..15 at the eight phase
 * boundaries of each k-tile selected by `mask` (bit 2i: own work of phase i done, bit 2i+1: the barrier behind it released; phases L01 M01
 * L23 M23): buffer[64 * group + 8 * (kt - 8) + point].  NULL switches it off (default).  tools/gemm_bf16_probe.py prints the timeline. */
int ytvln_gemm_bf16_probe(uint32_t* buffer, int block, int mask);
/* out[r][c] = bf16(x[r][c]), round to nearest even: network inputs that arrive as fp32 (the 2048-d region features, once per step) and
 * parameters that are not inside the optimizer's arenas yet (the steps before the first optimizer step). */
int ytvln_cast_f32_bf16(const float* x, int64_t ldx, int64_t rows, int cols, uint16_t* out, int64_t ldo, void* stream);

/* On-device batch preparation: the masking the reference applies per item on the host.
 *   ytvln_randomize_tokens  = randomize_tokens (utils/dataset/common.py:213-270, mask_action_rate = 0):  p = U[0,1) * mask;
 *     p >= 0.85: target = token, token = [MASK];  p >= 0.97: token = random id;  p >= 0.985: token = original;  targets -1 elsewhere.
 *   ytvln_randomize_regions = randomize_regions (common.py:272-300):  p >= 0.85: targets = probs, targets_mask = 1 (else 1/C, 0);
 *     p >= 0.865: the feature row is zeroed IN PLACE.
 *   Draws: explicit (`p`, `random_tokens`; pins the kernels bit-for-bit to the reference functions) or, when those are NULL, the
 *   Philox stream keyed by the device-resident (seed, counter) pair `rng` and `site` (the same state as the dropout kernels). */
int ytvln_randomize_tokens(const int64_t* tokens, const int64_t* mask, int64_t n, int vocab_size, int64_t mask_token_id,
                           const float* p, const int64_t* random_tokens, const int64_t* rng, int64_t site,
                           int64_t* tokens_out, int64_t* targets_out, void* stream);
int ytvln_randomize_regions(float* features, int64_t ldf, const float* probs, const int64_t* mask, int64_t rows, int F, int C,
                            const float* p, const int64_t* rng, int64_t site, float* targets, int64_t* targets_mask, void* stream);

/* out[j, :] = x[idx[j], :] (zeros where idx[j] < 0): row gather in front of the loss-aware prediction heads (only rows that
 * carry a masked-language / masked-vision target are decoded; "next" row of SURVEY.md section 8f).  Backward =
 * ytvln_scatter_add_rows_f32 with skip_idx = -1. */
int ytvln_gather_rows_f32(const float* x, int64_t ldx, const int64_t* idx, int R, int H, float* out, void* stream);

/* Fused (dropout ->) residual add -> LayerNorm (-> dropout).  BertLayerNorm, vilbert.py:213-217 (biased variance, eps
 * inside the sqrt) together with the dropout/add that always precedes it (:322-324, :365-367, :641-648) or follows it
 * (:254-255, :1367-1368).
 *   s = (p_pre > 0 ? dropout(x) : x) + (res ? res : 0);  y = gamma * (s - mean) * rstd + beta;  p_post: y = dropout(y)
 * s_out receives s (may alias x; may be NULL when not training), mean/rstd are [rows] (may be NULL). */
int ytvln_ln_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* s_out,
                     float* mean, float* rstd, int64_t rows, int H, float eps, float p_pre, float p_post,
                     const int64_t* rng, int64_t site, void* stream);

/* Backward of the above.  ds = dL/ds (gradient w.r.t. the residual input), dx = gradient w.r.t. x (written only when
 * p_pre > 0; otherwise dx == ds and dx may be NULL).  partial is [nblocks, 2, H] (dgamma then dbeta partial sums, to be
 * reduced with ytvln_colsum_f32); nblocks = ytvln_ln_bwd_blocks(rows). */
int ytvln_ln_bwd_blocks(int64_t rows);
int ytvln_ln_bwd_f32(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                     float* ds, float* dx, float* partial, int64_t rows, int H, float p_pre, float p_post,
                     const int64_t* rng, int64_t site, void* stream);

/* BertEmbeddings.forward, vilbert.py:240-256: s = word[ids] + pos[t] + type[tt]; y = dropout(LN(s)). rows = N*T. */
int ytvln_text_embed_fwd_f32(const int64_t* ids, const int64_t* type_ids, const float* word, const float* pos,
                             const float* type, const float* gamma, const float* beta, float* y, float* s_out,
                             float* mean, float* rstd, int64_t rows, int T, int H, float eps, float p_post,
                             const int64_t* rng, int64_t site, void* stream);

/* BertImageEmbeddings.forward after the 2048->Hv projection, vilbert.py:1361-1368:
 *   s = img + W5.loc[0:5] + b5 + W4.loc[5:9] + b4 + W2.loc[9:11] + b2 + E32[(int)loc[11]];  y = dropout(LN(s))
 * img is [rows,H] (already contains its own bias), loc is [rows,12]; W5 [H,5], W4 [H,4], W2 [H,2], E [32,H]. */
int ytvln_image_embed_fwd_f32(const float* img, const float* loc, const float* W5, const float* b5, const float* W4,
                              const float* b4, const float* W2, const float* b2, const float* E, const float* gamma,
                              const float* beta, float* y, float* s_out, float* mean, float* rstd, int64_t rows,
                              int H, float eps, float p_post, const int64_t* rng, int64_t site, void* stream);

/* dz = dy * act'(aux) elementwise (act = YTVLN_EPI_GELU: aux is the pre-activation; YTVLN_EPI_RELU: aux is the output). */
int ytvln_act_bwd_f32(const float* dy, const float* aux, float* dz, int64_t n, int act, void* stream);

/* y = x * keep / (1 - p): nn.Dropout forward and (applied to dy) backward (lily.py:100). */
int ytvln_dropout_f32(const float* x, float* y, int64_t n, float p, const int64_t* rng, int64_t site, void* stream);

/* bf16-resident forms of the five kernels above (BASELINE configs[4]): rows of x / res / y / s_out / dy / s / ds / dx are bf16 (8-byte aligned),
 * the arithmetic is the fp32 one (values widened on load, rounded to nearest even on store), statistics, gamma / beta, their gradient partials and
 * the embedding tables stay fp32; the dropout masks are the SAME as in the fp32 kernels (one Philox draw per group of four elements).
 * ytvln_ln_bwd_bf16: ds / dx may be NULL when the caller only wants ds_f32, an fp32 copy of ds for the embedding-table gradient kernels. */
int ytvln_ln_fwd_bf16(const uint16_t* x, const uint16_t* res, const float* gamma, const float* beta, uint16_t* y, uint16_t* s_out, float* mean,
                      float* rstd, int64_t rows, int H, float eps, float p_pre, float p_post, const int64_t* rng, int64_t site, void* stream);
int ytvln_ln_bwd_bf16(const uint16_t* dy, const uint16_t* s, const float* mean, const float* rstd, const float* gamma, uint16_t* ds, uint16_t* dx,
                      float* ds_f32, float* partial, int64_t rows, int H, float p_pre, float p_post, const int64_t* rng, int64_t site, void* stream);
int ytvln_text_embed_fwd_bf16(const int64_t* ids, const int64_t* type_ids, const float* word, const float* pos, const float* type,
                              const float* gamma, const float* beta, uint16_t* y, uint16_t* s_out, float* mean, float* rstd, int64_t rows,
                              int T, int H, float eps, float p_post, const int64_t* rng, int64_t site, void* stream);
int ytvln_image_embed_fwd_bf16(const uint16_t* img, const float* loc, const float* W5, const float* b5, const float* W4, const float* b4,
                               const float* W2, const float* b2, const float* E, const float* gamma, const float* beta, uint16_t* y,
                               uint16_t* s_out, float* mean, float* rstd, int64_t rows, int H, float eps, float p_post, const int64_t* rng,
                               int64_t site, void* stream);
int ytvln_act_bwd_bf16(const uint16_t* dy, const uint16_t* aux, uint16_t* dz, int64_t n, int act, void* stream);

/* Fused multi-head attention on the fp32 matrix cores, flash-style (scores never reach HBM).  One entry point serves
 * BertSelfAttention (vilbert.py:284-311), BertImageSelfAttention (:413-440) and both directions of BertBiAttention
 * (:577-616):   ctx[n, i, h*d:(h+1)*d] = softmax_j(q_i.k_j * scale + mask[n, j]) (dropout) . v_j
 *   q: rows n*Tq+i, head h at columns h*d.., leading dimension ldq (so packed QKV projections are consumed in place);
 *   k, v likewise with Tk rows per n; mask: additive [N, Tk] (0 / -10000, vilbert.py:1282,1287) or NULL;
 *   lse [N, heads, Tq] receives log-sum-exp of the scaled+masked scores (saved for backward).  d in {32, 64, 128}. */
int ytvln_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                       const float* mask, float* ctx, int64_t ldo, float* lse, int N, int heads, int Tq, int Tk,
                       int d, float scale, float p_drop, const int64_t* rng, int64_t site, void* stream);

/* Backward: delta [N,heads,Tq] is scratch (row sums of dctx*ctx).  dq/dk/dv use the same strided layout as q/k/v and
 * are overwritten. */
int ytvln_attn_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                       const float* mask, const float* ctx, const float* dctx, int64_t ldo, const float* lse,
                       float* delta, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, int N,
                       int heads, int Tq, int Tk, int d, float scale, float p_drop, const int64_t* rng, int64_t site,
                       void* stream);

/* Both directions of BertBiAttention (vilbert.py:552-618) in ONE launch per kernel: text queries over region keys/values and region
 * queries over text keys/values are independent problems of complementary shape (T x R and R x T); one grid holds the workgroups of
 * both, so the slots one direction's last partial round would leave idle are filled by the other.  Each problem is described like the
 * arguments of ytvln_attn_fwd_f32 / ytvln_attn_bwd_f32 (forward reads q,k,v,mask and writes ctx,lse; backward reads
 * q,k,v,mask,ctx_in,dctx,lse_in and writes delta,dq,dk,dv); N, heads, d, scale and rng are shared. */
typedef struct ytvln_attn_problem {
    const float *q, *k, *v, *mask;
    const float *ctx_in, *dctx, *lse_in;
    float *ctx, *lse, *delta, *dq, *dk, *dv;
    int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int32_t Tq, Tk;
    float p_drop;
    int32_t reserved;
    int64_t site;
    void* keep;      /* bf16 entry points with p_drop > 0: the keep decisions of the dropout, ytvln_attn_keep_bytes(N, heads, Tq, Tk) bytes, 128-byte
                        aligned -- WRITTEN by ytvln_attn_fwd_bf16 (16 64-bit lane masks per 32x32 block of scores), READ by ytvln_attn_bwd_bf16 of the
                        same problem; in a two-problem launch BOTH records need one as soon as either has p_drop > 0; ignored by the fp32 entry points (they
                        regenerate the hash) and by launches without dropout (may be NULL) */
} ytvln_attn_problem;
/* sizeof(ytvln_attn_problem) as the LIBRARY was built: a binding asserts it equals its own record size at load time */
int64_t ytvln_attn_problem_size(void);
int ytvln_attn_fwd_pair(const ytvln_attn_problem* a, const ytvln_attn_problem* b, int N, int heads, int d, float scale,
                        const int64_t* rng, void* stream);
int ytvln_attn_bwd_pair(const ytvln_attn_problem* a, const ytvln_attn_problem* b, int N, int heads, int d, float scale,
                        const int64_t* rng, void* stream);
/* The same attention for the bf16-resident path (BASELINE configs[4]): in the problem records q, k, v, ctx, ctx_in, dctx, dq, dk, dv point to
 * BF16 tensors (leading dimensions in elements), mask / lse / lse_in / delta stay fp32; every contraction runs on v_mfma_f32_32x32x16_bf16
 * with fp32 accumulation and fp32 softmax, the transposed operands (V^T.P^T, K^T.dS^T, Q^T.dS, dO^T.P) are gathered from the row-major LDS
 * tiles by ds_read_b64_tr_b16.  `b` may be NULL (one problem: self-attention) or the second direction of BertBiAttention.  Head dimension
 * 64 or 128. */
int64_t ytvln_attn_keep_bytes(int N, int heads, int Tq, int Tk);
int ytvln_attn_fwd_bf16(const ytvln_attn_problem* a, const ytvln_attn_problem* b, int N, int heads, int d, float scale,
                        const int64_t* rng, void* stream);
int ytvln_attn_bwd_bf16(const ytvln_attn_problem* a, const ytvln_attn_problem* b, int N, int heads, int d, float scale,
                        const int64_t* rng, void* stream);

/* probs[n,h,i,j] = exp(q_i.k_j*scale + mask - lse): the attention_probs tensor the reference returns when
 * output_all_attention_masks=True (vilbert.py:300, 311).  Diagnostic path, not on the training step. */
int ytvln_attn_probs_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* mask, const float* lse,
                         float* probs, int N, int heads, int Tq, int Tk, int d, float scale, void* stream);

/* F.cross_entropy(logits, target, ignore_index) (utils_init.py:133-135, 141) -- also with -inf padded logits.
 * fwd: row_lse [M], out[0] = mean loss over non-ignored rows (NaN when none, like the reference), out[1] = their count.
 * bwd: dlogits = (softmax - onehot) * gout[0] / count for valid rows, 0 otherwise. */
int ytvln_ce_fwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index, float* row_lse,
                     float* row_loss, float* out, int M, int V, void* stream);
int ytvln_ce_bwd_f32(const float* logits, int64_t ld, const int64_t* target, int64_t ignore_index, const float* row_lse,
                     const float* out, const float* gout, float* dlogits, int64_t ldd, int M, int V, void* stream);
/* bf16-resident path (BASELINE configs[4]): the same two kernels on BF16 logits (what the decoders write there); the loss, row_lse and out
 * stay fp32; the gradient is bf16 and ZEROS are written to its padding columns [V, ldd), so the buffer feeds ytvln_gemm_bf16 as a zero-padded
 * operand (YTVLN_GEMM_A_ZERO_PADDED) as it stands */
int ytvln_ce_fwd_bf16(const uint16_t* logits, int64_t ld, const int64_t* target, int64_t ignore_index, float* row_lse,
                      float* row_loss, float* out, int M, int V, void* stream);
int ytvln_ce_bwd_bf16(const uint16_t* logits, int64_t ld, const int64_t* target, int64_t ignore_index, const float* row_lse,
                      const float* out, const float* gout, uint16_t* dlogits, int64_t ldd, int M, int V, void* stream);

/* Masked KL of utils_init.py:117-128: sum_rows mask * sum_c t*(log t - log_softmax(pred)) / max(1, sum mask). */
int ytvln_kl_fwd_f32(const float* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask,
                     float* row_lse, float* row_loss, float* out, int M, int C, void* stream);
int ytvln_kl_bwd_f32(const float* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask,
                     const float* row_lse, const float* out, const float* gout, float* dpred, int64_t ldd, int M,
                     int C, void* stream);
/* bf16 predictions in, bf16 gradient + zeroed padding out, as ytvln_ce_*_bf16 (targets stay fp32) */
int ytvln_kl_fwd_bf16(const uint16_t* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask, float* row_lse,
                      float* row_loss, float* out, int M, int C, void* stream);
int ytvln_kl_bwd_bf16(const uint16_t* pred, int64_t ld, const float* target, int64_t ldt, const int64_t* mask, const float* row_lse,
                      const float* out, const float* gout, uint16_t* dpred, int64_t ldd, int M, int C, void* stream);

/* F.binary_cross_entropy_with_logits(x, t, pos_weight) with mean reduction (utils_init.py:143, 160-161). n <= 65536.
 * pos_weight is a DEVICE scalar or NULL.  bwd writes dx. */
int ytvln_bce_fwd_f32(const float* x, const float* t, const float* pos_weight, float* out, int n, void* stream);
int ytvln_bce_bwd_f32(const float* x, const float* t, const float* pos_weight, const float* gout, float* dx, int n,
                      void* stream);

/* Fused AdamW of vilbert/optimization.py:141-187 over flat arenas.  chunks is a DEVICE array of nchunks records
 * {int64 offset, int64 length, float weight_decay, float pad} (24 bytes); hyper is a DEVICE array
 * {beta1, beta2, eps, step_size = lr*sqrt(1-b2^t)/(1-b1^t), lr} of floats (updated by the host per step; graph-safe).
 *   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= step_size * m/(sqrt(v)+eps);  p -= lr*wd*p   (decay after update)
 * grad_scale multiplies g on the fly (1/world_size after a summed all-reduce). */
int ytvln_adamw_f32(float* p, const float* g, float* m, float* v, const void* chunks, int nchunks, const float* hyper,
                    float grad_scale, void* stream);
/* the same step, ALSO writing bf16(p) (round to nearest even) at the same offsets of a bf16 arena: the weight operands of the bf16-resident
 * path are refreshed by the optimizer step itself (+2 bytes per parameter, no cast pass) */
int ytvln_adamw_f32_bf16copy(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, const void* chunks, int nchunks,
                             const float* hyper, float grad_scale, void* stream);

/* ---- data-parallel gradient exchange: RCCL over xGMI ------------------------------------------------------------------------
 * Replaces DistributedDataParallel over NCCL (utils/distributed.py:63-104: init_process_group("nccl") + DDP's bucketed all-reduce).
 * librccl is resolved with dlopen at run time (no link dependency): an explicit path, else a librccl already mapped into the
 * process (PyTorch's own copy -- same HIP runtime, so streams and device pointers interoperate), else the loader's default.
 * A communicator is an opaque handle owned by the caller (the one piece of state that outlives a call); collectives are
 * asynchronous on the stream they are given and operate IN PLACE on device memory; nothing here synchronises or allocates
 * device memory.  The 128-byte unique id is created on rank 0 and carried to the other ranks by the host (env:// store). */
enum { YTVLN_DT_F32 = 0, YTVLN_DT_F64 = 1, YTVLN_DT_BF16 = 2, YTVLN_DT_I64 = 3, YTVLN_DT_U8 = 4 };
enum { YTVLN_RED_SUM = 0, YTVLN_RED_MAX = 1, YTVLN_RED_MIN = 2 };
#define YTVLN_RCCL_UNIQUE_ID_BYTES 128
int ytvln_rccl_load(const char* path);                 /* path may be NULL / "" (see the resolution order above)            */
const char* ytvln_rccl_library_path(void);            /* file the nccl* symbols were bound from ("" before a load)           */
int ytvln_rccl_version(int* version);                  /* NCCL_VERSION_CODE of the loaded library                              */
int ytvln_rccl_unique_id(void* id_out, int64_t bytes); /* ncclGetUniqueId; bytes must be YTVLN_RCCL_UNIQUE_ID_BYTES           */
/* ncclCommInitRank (collective over all `world` ranks; blocks until they all arrive).  device >= 0: hipSetDevice first. */
int ytvln_rccl_init(void** comm_out, const void* id, int64_t id_bytes, int rank, int world, int device);
/* in-place all-reduce of `count` elements; dtype / op from the enums above */
int ytvln_rccl_allreduce(void* comm, void* buf, int64_t count, int dtype, int op, void* stream);
/* SUM all-reduce of `nslices` contiguous slices base[offsets[i] : offsets[i]+counts[i]] (HOST arrays) of one flat fp32 gradient
 * arena as ONE RCCL group: the buckets of an optimizer step without a host round trip between them. */
int ytvln_rccl_allreduce_slices_f32(void* comm, float* base, const int64_t* offsets, const int64_t* counts, int nslices,
                                    void* stream);
/* the same for an arena of any element type of the enum above (offsets / counts in ELEMENTS): the bf16 gradient exchange of the bf16-resident
 * path (BASELINE configs[4]) moves half the bytes of the fp32 one over xGMI */
int ytvln_rccl_allreduce_slices(void* comm, void* base, int dtype, const int64_t* offsets, const int64_t* counts, int nslices,
                                void* stream);
/* in-place byte broadcast from `root` (DDP's rank-0 weight broadcast at wrap time) */
int ytvln_rccl_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream);
int ytvln_rccl_async_error(void* comm);                /* 0 = healthy; negative + ytvln_last_error() otherwise                 */
int ytvln_rccl_destroy(void* comm);                    /* ncclCommDestroy; NULL is a no-op                                     */

#ifdef __cplusplus
}
#endif
#endif /* YTVLN_H */
