/* include/ps_hip.h — C-ABI of the MI355X (gfx950) backend for PowerServe's ggml decode hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry
 * point names the reference interface it replaces (paths relative to the PowerServe tree).
 *
 *   reference                                             this library
 *   ---------------------------------------------------   ---------------------------------------------
 *   powerserve_compute_forward_mul_mat  (ggml.h:767)      ps_hip_mul_mat
 *   powerserve_compute_forward_add      (ggml.h:774)      ps_hip_add
 *   powerserve_compute_forward_rms_norm (ggml.h:787)      ps_hip_rms_norm
 *   powerserve_compute_forward_rope     (ggml.h:795)      ps_hip_rope
 *   powerserve_compute_forward_dup      (ggml.h:804)      ps_hip_dup
 *   powerserve_compute_forward_softmax_ext (ggml.h:810)   ps_hip_softmax_ext
 *   powerserve_compute_forward_soft_max (ggml.h:781)      ps_hip_soft_max
 *   powerserve_get_vec_dot_type         (ggml.h:765)      ps_hip_vec_dot_type
 *   GGMLBackend::silu_hadamard (backend/ggml/ggml.cpp:115)        ps_hip_silu_hadamard
 *   GGMLBackend::get_embedding (backend/ggml/ggml_wrapper.cpp:181) ps_hip_get_embedding
 *   Executor GET_MASK          (executor/executor.cpp:210-224)    ps_hip_get_mask
 *   quantize_row_q8_0 / q8_K   (ggml-quants.c:887, :3849)         ps_hip_quantize_act
 *   GGMLBackend::plan + fused decode (backend/ggml/ggml.cpp:30)   ps_hip_model_* (one call per forward)
 *
 * Conventions
 *   - ps_tensor mirrors ggml_tensor's (type, ne[4], nb[4] in BYTES, data) as built by convert_to_ggml
 *     (backend/ggml/ggml.hpp:87-96).  `data` is a DEVICE pointer, except for quantized weights where it is
 *     the ps_weight handle returned by ps_hip_weight_upload (the device copy is repacked).
 *   - every function returns 0 on success, non-zero on failure; ps_hip_last_error(ctx) gives the message.
 *     Nothing throws across this boundary; the C++ façade converts failures into the reference's
 *     abort/throw behaviour (core/logger.hpp:56-82).
 *   - one HIP stream per ctx; ops are asynchronous in stream order; ps_hip_sync waits.  A ctx is not
 *     thread-safe (same as one GGMLBackend).
 */
#ifndef PS_HIP_H
#define PS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_HIP_ABI_VERSION 2
/* return codes: 0 ok; 1 a HIP runtime error; 2 a refused call (bad argument, unsupported shape ...); 3 see ps_hip_model_sync_check */
#define PS_HIP_ATTN_TIMEOUT 3

/* ggml_type values (libs/ggml/include/ggml.h:361-398); Q4_K/Q5_K/Q6_K extend PowerServe's DataType enum
 * (core/data_type.hpp:24-35) which stops at Q8_0. */
enum ps_dtype {
    PS_F32  = 0,
    PS_F16  = 1,
    PS_Q4_0 = 2,
    PS_Q8_0 = 8,
    PS_Q4_K = 12,
    PS_Q5_K = 13,
    PS_Q6_K = 14,
    PS_Q8_K = 15,
    PS_I32  = 26,
};

typedef struct ps_hip_ctx ps_hip_ctx;
typedef struct ps_weight ps_weight;   /* device-resident quantized weight matrix [K, N] */
typedef struct ps_hip_model ps_hip_model;
typedef struct ps_hip_graph ps_hip_graph;

typedef struct {
    int32_t dtype;
    int32_t _pad;
    int64_t ne[4];  /* elements per dim, dim0 fastest */
    uint64_t nb[4]; /* strides in bytes */
    void *data;
} ps_tensor;

/* rope_compute_params (ggml.h:651-661) */
typedef struct {
    int32_t n_dims, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
    int32_t mode; /* rope_type: 0 adjacent pairs, 2 NEOX */
} ps_rope_params;

/* ModelConfig::LLMConfig (core/config.hpp:86-109) */
typedef struct {
    uint32_t dim, hidden_dim, n_layers, n_heads, n_kv_heads, seq_len, vocab_size, kv_dim, head_size;
    float norm_eps;
    ps_rope_params rope;
} ps_llm_config;

/* ------------------------------------------------------------------ context, memory, timing */
int ps_hip_abi_version(void);
/* Which build of the reference this library's fp32 arithmetic follows bit for bit.  0 (lib/libps_hip.so, the default and what every "bit-exact"
 * in this repository refers to): the reference compiled with -ffp-contract=off -- each operation rounds where the C source rounds.  1
 * (lib/libps_hip_contract.so, built with -DPS_CONTRACT by powerserve_amd/build.py): the reference as its own CMake compiles it on an FMA machine
 * (no contraction flag: GCC's default -ffp-contract=fast; CMakeLists.txt:24-33, libs/ggml/src/CMakeLists.txt:1173), which fuses the RoPE rotation
 * (ggml.c:15455-15475) and the n % 32 leftovers of ggml_vec_dot_f32 (ggml.c:2123-2125); Q5_K (a third fused site, ggml-quants.c:8411) is
 * refused by that build. */
int ps_hip_build_contract(void);
int ps_hip_device_count(void);
int ps_hip_create(int device, ps_hip_ctx **out);
void ps_hip_destroy(ps_hip_ctx *ctx);
const char *ps_hip_last_error(const ps_hip_ctx *ctx);
int ps_hip_device_name(const ps_hip_ctx *ctx, char *buf, size_t cap);
int ps_hip_malloc(ps_hip_ctx *ctx, size_t bytes, void **dptr);
int ps_hip_free(ps_hip_ctx *ctx, void *dptr);
int ps_hip_memcpy_h2d(ps_hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int ps_hip_memcpy_d2h(ps_hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int ps_hip_memset(ps_hip_ctx *ctx, void *dst_dev, int value, size_t bytes);
int ps_hip_sync(ps_hip_ctx *ctx);
void *ps_hip_stream(ps_hip_ctx *ctx); /* hipStream_t */
/* HIP events on the ctx stream (bench.py measures kernel time with these) */
int ps_hip_event_create(ps_hip_ctx *ctx, void **ev);
int ps_hip_event_record(ps_hip_ctx *ctx, void *ev);
int ps_hip_event_elapsed_ms(ps_hip_ctx *ctx, void *ev_start, void *ev_stop, float *ms); /* syncs ev_stop */
int ps_hip_event_destroy(ps_hip_ctx *ctx, void *ev);

/* ------------------------------------------------------------------ weights */
/* host_blocks: N rows of GGUF blocks of `dtype` (row stride = ggml_row_size(dtype, K)), or F32 rows.
 * The device copy is repacked into a structure-of-arrays layout private to the backend. */
int ps_hip_weight_upload(ps_hip_ctx *ctx, int dtype, const void *host_blocks, int64_t K, int64_t N, ps_weight **out);
void ps_hip_weight_free(ps_hip_ctx *ctx, ps_weight *w);
/* GGUF bytes of the matrix = N * ggml_row_size(dtype, K): the roofline's algorithmic byte count */
uint64_t ps_hip_weight_gguf_bytes(const ps_weight *w);
int ps_hip_weight_dtype(const ps_weight *w);

/* ------------------------------------------------------------------ op level (reference-shaped) */
int ps_hip_vec_dot_type(int dtype);
size_t ps_hip_row_size(int dtype, int64_t k);
/* x: [K, rows] F32 contiguous (device) -> out: rows x GGUF-layout block_q8_0 / block_q8_K (device).
 * Bit-exact with quantize_row_q8_0 (AVX2 branch) / quantize_row_q8_K. */
int ps_hip_quantize_act(ps_hip_ctx *ctx, int vdt, const float *x, int64_t K, int64_t rows, void *out_blocks);
/* dst[ne01, ne11, ne12] = src0 x src1.  src0: quantized weight handle (2-D) or F32 tensor with arbitrary
 * nb[1..3] (K/V-cache views, GQA broadcast over dim 2); src1: F32, nb[0]==4; dst: F32 contiguous rows. */
int ps_hip_mul_mat(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *src0, const ps_tensor *src1);
int ps_hip_rms_norm(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *src, const ps_tensor *weight, float eps);
/* pos: HOST int32[n_pos] (the reference passes a std::vector<int>, executor.cpp:123-128) */
int ps_hip_rope(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *src, const int32_t *pos, int n_pos,
                const ps_rope_params *rp);
int ps_hip_softmax_ext(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *src, const ps_tensor *mask,
                       float scale, float max_bias);
/* GGMLBackend::softmax (backend/ggml/ggml_wrapper.cpp:57-69) -> powerserve_compute_forward_soft_max (ggml.c:15060-15089): the same
 * row soft-max with scale 1, no mask, max_bias 0 (it sets exactly those op_params and calls ggml_compute_forward_soft_max_f32). */
int ps_hip_soft_max(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *src);
int ps_hip_add(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *a, const ps_tensor *b);
int ps_hip_dup(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *src); /* COPY and CONT */
int ps_hip_silu_hadamard(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *gate, const ps_tensor *up);
/* weight->data: ps_weight handle (quantized or F32 table [dim, vocab]); tokens: HOST int32[n] */
int ps_hip_get_embedding(ps_hip_ctx *ctx, const ps_tensor *dst, const ps_tensor *weight, const int32_t *tokens,
                         int n);
/* dst [n_kv, bs] F32: (j <= pos[i]) ? 0 : -inf  — or, with tree != NULL (bs x bs bytes, row i = which batch
 * tokens token i may attend to), the speculative-decode tree mask over the last bs cache slots. */
int ps_hip_get_mask(ps_hip_ctx *ctx, const ps_tensor *dst, const int32_t *pos, int n_pos, const uint8_t *tree);
/* rows of `src` ([n, rows] F32) -> int32 index of the first maximum per row, written to out_dev */
int ps_hip_argmax(ps_hip_ctx *ctx, const float *src, int64_t n, int64_t rows, int32_t *out_dev);

/* ------------------------------------------------------------------ whole-model fast path
 * What HIPBackend::plan() lowers the reference's 28-op layer sequence to: fused kernels, a persistent
 * arena, a device-resident FP32 KV cache (K [n_ctx][kv_dim], V [kv_dim][n_ctx] — the reference's layout,
 * backend/ggml/ggml_kv_cache.cpp:48-57, model/module/norm_attention.cpp:82-104) and a captured hipGraph
 * for the single-token step. */
typedef struct {
    ps_llm_config cfg;
    int32_t is_qwen2; /* QKV bias (model/qwen2/qwen2_model.cpp:75) */
    int32_t max_batch; /* largest forward batch (prefill chunk) */
    const ps_weight *token_embd, *output; /* output == NULL -> tied lm_head (weights.hpp:67-68) */
    const float *output_norm;             /* device F32 [dim] */
    /* per layer arrays of n_layers entries */
    const float *const *attn_norm, *const *ffn_norm;
    const ps_weight *const *attn_q, *const *attn_k, *const *attn_v, *const *attn_output;
    const ps_weight *const *ffn_gate, *const *ffn_up, *const *ffn_down;
    const float *const *attn_q_bias, *const *attn_k_bias, *const *attn_v_bias; /* NULL unless is_qwen2 */
} ps_model_desc;

int ps_hip_model_create(ps_hip_ctx *ctx, const ps_model_desc *desc, ps_hip_model **out);
void ps_hip_model_destroy(ps_hip_model *m);
/* KVCacheInterface bookkeeping (core/kv_cache.hpp:97-163) */
size_t ps_hip_model_kv_position(const ps_hip_model *m);
int ps_hip_model_max_batch(const ps_hip_model *m); /* widest forward the model's buffers hold (ps_model_desc::max_batch) */
int ps_hip_model_kv_truncate(ps_hip_model *m, size_t n_tokens);
int ps_hip_model_kv_advance(ps_hip_model *m, size_t n_tokens); /* after an op-by-op forward through the ps_hip_* operators */
int ps_hip_model_kv_rollback(ps_hip_model *m, size_t n_tokens);
int ps_hip_model_kv_move(ps_hip_model *m, size_t dst_index, size_t src_index);
/* The rest of KVCacheInterface (core/kv_cache.hpp:120-162).  The reference keeps a forward's K / V rows in a per-batch staging area
 * (GGMLKV::chunk.current_k / current_v) and copies them into the cache afterwards; here every forward writes its rows straight into
 * the cache slots kv_position + i (norm_attention.cpp:82-104 does the same through VIEW + COPY on the ggml path), so "token i of the
 * last batch" IS cache slot kv_position + i:
 *   copy(dst_cache_index, src_token_index)  = move(dst_cache_index, kv_position + src_token_index)
 *   save_tokens(n)                          = nothing left to copy (checks kv_position + n <= n_ctx like the reference's assert)
 *   unmask_tokens(n)                        = slots kv_position .. kv_position + n - 1 visible again; the position is not modified
 *   append_tokens(n)                        = save_tokens + unmask_tokens + advance_tokens; returns 0 and the OLD position in *old_position
 *                                             (may be NULL) */
int ps_hip_model_kv_copy(ps_hip_model *m, size_t dst_cache_index, size_t src_token_index);
int ps_hip_model_kv_save_tokens(ps_hip_model *m, size_t n_tokens);
int ps_hip_model_kv_unmask_tokens(ps_hip_model *m, size_t n_tokens);
int ps_hip_model_kv_append_tokens(ps_hip_model *m, size_t n_tokens, size_t *old_position);
/* One Model::forward (model/llama/llama_model.cpp:52-117).  tokens/pos: HOST arrays of n entries,
 * positions consecutive from pos[0].  tree (may be NULL): bs x bs attention mask among the batch tokens
 * (speculative tree verify).  lm_head != 0: logits for all n tokens are left in the model's device buffer
 * (ps_hip_model_logits) and their arg-max ids are written to argmax_host (may be NULL).  Advances the KV
 * position by n (m_kv->advance, llama_model.cpp:109). */
int ps_hip_model_forward(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *pos, const uint8_t *tree,
                         int lm_head, int32_t *argmax_host);
/* The same launches as ps_hip_model_forward, for a graph that HIPBackend::plan lowered (src/executor/executor.cpp:47-49,79:
 * the whole op vector goes to the backend's plan() before it runs): enqueues only -- no host sync, the KV position is left
 * to the caller (LlamaModel::forward advances it after Executor::run, llama_model.cpp:109).  The result counts once
 * ps_hip_model_kv_advance or ps_hip_model_sync_check has returned 0. */
int ps_hip_model_forward_lowered(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *pos, const uint8_t *tree, int lm_head);
/* Waits for a pending ps_hip_model_forward_lowered and looks at the one-launch attention's time-out flag (k_attn.hip: its
 * workgroups wait for each other inside the launch, bounded).  0: the result is valid.  PS_HIP_ATTN_TIMEOUT: it is not -- the model
 * has switched to the two-launch attention (sticky, see ps_hip_model_set_mode) and the caller runs the forward again, once (nothing
 * was advanced; behind the switch a forward cannot time out).  1: a HIP error -- not a time-out, do not retry.
 * ps_hip_model_kv_advance calls it, and so does every other model entry point before it enqueues anything, so an unconsumed lowered
 * forward's time-out is reported to the caller that still holds its inputs instead of being pinned on the next forward. */
int ps_hip_model_sync_check(ps_hip_model *m);
/* ModelTokenIterator's prefill loop (src/model/model.hpp:147-163: forward(chunk, lm_head = false) + advance, chunk after chunk of
 * `chunk` tokens = hparams batch_size) for tokens appended at the current cache position -- bit-identical to calling
 * ps_hip_model_forward per chunk, but up to max_batch / chunk chunks share one launch sequence: the mat-muls take all their columns at
 * once, only the attention is evaluated per reference chunk (its sums depend on where a chunk ends). */
int ps_hip_model_prefill(ps_hip_model *m, const int32_t *tokens, int n, int chunk);
/* Token-tree forward (speculative verify / draft, src/speculative/token_tree.cpp): the n tokens are appended at the
 * cache slots kv_position .. kv_position+n-1, column i is rotated with RoPE position rope_pos[i] (its depth in the tree,
 * not its slot) and sees the cached prefix (minus slots hidden with ps_hip_model_kv_mask) plus the batch columns j with
 * tree[i*n + j] != 0 (NULL: causal).  advance = 0 leaves kv_position where it was: the caller keeps the accepted path
 * with ps_hip_model_kv_move(dst, src) + ps_hip_model_kv_advance(1) per node, as TokenTree::verify does. */
int ps_hip_model_forward_tree(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *rope_pos, const uint8_t *tree,
                              int lm_head, int32_t *argmax_host, int advance);
/* KVCacheInterface::mask / unmask (core/kv_cache.hpp:97-163): hide / show one cached slot in every later forward */
int ps_hip_model_kv_mask(ps_hip_model *m, size_t index, int visible);
/* Greedy single-token steps, the decode hot loop (model/model.hpp:170-183): feeds `token` at the current
 * KV position, then its own arg-max, `steps` times, without host round trips (hipGraph replay).  out_ids
 * HOST int32[steps]. */
int ps_hip_model_decode_greedy(ps_hip_model *m, int32_t token, int steps, int32_t *out_ids);
const float *ps_hip_model_logits(const ps_hip_model *m); /* device [max_batch][vocab] */
/* Arg-max ids of the most recent forward with lm_head (first maximum per token, what greedy_sample / TopK(1) returns: src/sampler/prob_array.cpp:65-67,
 * sampler.cpp:39-56), computed on the device behind the lm_head: 4 bytes per token come to the host instead of vocab x 4 (513 KB for 128 256 logits,
 * src/model/model.hpp:170-183).  For a lowered forward call it after ps_hip_model_sync_check / ps_hip_model_kv_advance: the ids have then already arrived
 * in pinned memory behind the forward's launches and the call does not wait on the stream again (one wait per decode step on the op-API path). */
int ps_hip_model_argmax(ps_hip_model *m, int n, int32_t *ids_host);
/* diagnostics: device scratch tensors of the most recent forward, last layer (0 x, 1 q, 2 att, 3 ffn hidden, 4 scores) */
const float *ps_hip_model_scratch(const ps_hip_model *m, int which);
const float *ps_hip_model_k_cache(const ps_hip_model *m, int layer);
const float *ps_hip_model_v_cache(const ps_hip_model *m, int layer);
/* per-forward accounting for the roofline: GGUF bytes of all mat-mul weights streamed by one token */
uint64_t ps_hip_model_weight_bytes_per_token(const ps_hip_model *m);
/* Measurement helper (bench.py roofline): replays mat-vec launches of one single-token forward exactly as the decode
 * step issues them (same kernels and fused prologues, every layer's own weights) `reps` times between HIP events on
 * the ctx stream -> seq_ms per token.  which = 0: all quantized mat-vecs; 1: the gate/up launch of every layer only.
* (which: 0 every quantized mat-vec of a token, 1 gate/up, 2 QKV, 3 O, 4 down, 5 lm_head)
 * null_ms = the same number of empty launches (launch-boundary cost); n_launches = launches per token. */
int ps_hip_model_bench_gemv(ps_hip_model *m, int reps, int which, double *seq_ms, double *null_ms, int *n_launches);
/* the same replay with bs activation columns (a prefill chunk's mat-muls with their quantizer launches) */
int ps_hip_model_bench_matmul(ps_hip_model *m, int reps, int which, int bs, double *seq_ms, double *null_ms, int *n_launches);
/* Diagnostic: in-kernel timeline of the decode mat-vec (tools/gpu_timeline.py).  With host_out == NULL, arm
 * (key >= 0: record launches with epilogue*4 + prologue == key; key < 0: disarm).  With host_out != NULL, copy
 * the last recorded launch: n_words uint64 = [workgroup][role 0 producer wave 0 / 1 chain wave][32 events],
 * shader-clock ticks (s_memtime).  key = k1 + 100 * (k2 + 1) records a second launch family into a second block of
 * the same size (kernel-boundary gaps). */
/* diagnostics: the kernel (rocprofv3's template name) the most recent quantized mat-mul launch of this process went to */
const char *ps_hip_last_matmul_kernel(void);
int ps_hip_debug_timeline(ps_hip_ctx *ctx, int key, uint64_t *host_out, int n_words);
/* Diagnostic: tunables of the library (process-wide).  key 1: wave configuration of the decode mat-vec (k_gemv4.hip,
 * tools/g4_variants.py); key 3: the narrow-batch Q4_K mat-mul for few row tiles (k_gemm4k.hip: 0 = gemm4k_par_kernel off (the wave-per-tile / staged forms take the launch: A/B only),
 * 4 / 8 = gemm4k_par_kernel with that many waves per tile, 1 = by tile count, the default); key 4: the fp16 perf mode's GEMM (perf16.hip:
 * 0 = by shape, 1 = 128-token tiles, 2 = 256 x 256 tiles); key 5: the next `value` single-token forwards on the one-launch attention
 * report a time-out of its score exchange (tests of the retry paths: forward, forward_tree, prefill tail, decode_greedy, lowered forward +
 * kv_advance); key 6: column blocks per XCD of the wide Q4_K / Q5_K mat-mul's item order (k_gemm4k.hip g4k_item: 0 = round 2's order, default 4;
 * PS_G4K_CBX); key 7: 1 = the fused Q / K / V + attention launch (k_qkvattn.hip) wherever it is covered -- by default only where its grid fills three quarters
 * of the chip (head size 128 with 8 kv heads), the head-size-64 instance otherwise runs under test only; key 9: the single-token attention kernels read the cached K rows / V channels
 * with plain (0) or non-temporal (1) loads whatever the cache's size, -1 (default): by the rule of csrc/model.hip (a cache-policy hint: no result bit depends on it; PS_KV_STREAM).
 * Returns non-zero for an unknown key.  (A captured single-token step keeps the launch plan it was captured with: ps_hip_model_set_mode drops it.) */
int ps_hip_debug_set(int key, int value);
/* Diagnostic: one GEMM shape of the fp16 perf mode on synthetic operands (tools/f16_gemm_bench.py): out[M][N] = x[M][K] . W[N][K]^T, timed over
 * `reps` launches (with beta: out = beta out + ...), max |difference| to a k-ordered fp32 reference (beta != 0: of a second launch on top of
 * the first result against (1 + beta) x the reference). */
int ps_hip_debug_f16_gemm(ps_hip_ctx *ctx, int M, int64_t N, int64_t K, int reps, float beta, double *us_per_launch, double *max_abs_err);
/* bit 0: 0 = hipGraph replay of the decode step (default), 1 = eager launches (rocprofv3 needs them);
 * bits 1, 2: unused (round 1 / 2 experiments, removed);
 * bit 3: 1 = fp16-KV decode mode — NOT bit-exact: K and V are mirrored in fp16 as they are appended and the
 * single-token attention reads only the mirrors (half the KV bytes) with a split-KV online soft-max; prefill, batches and
 * tree verify keep reading the FP32 caches.  Must be switched on while the cache is empty (position 0).;
 * bit 4: 1 = single-token attention as TWO launches (scores, then soft-max + V.p) instead of the one-launch form
 * (attn_decode2_kernel: scores exchanged inside the launch; it needs every workgroup of its grid resident, its wait is bounded,
 * and a wait that gives up switches this bit on: the forward that hit it runs again on the two launches -- ps_hip_model_forward,
 * _forward_tree, _prefill and _decode_greedy do that themselves, a lowered forward reports it through ps_hip_model_sync_check /
 * ps_hip_model_kv_advance).  Same results bit for bit.  Sticky after a time-out: a later set_mode without bit 4 keeps the two launches;
 * bit 6 (write-only): re-arm the one-launch attention after a time-out (clears the sticky state, then the other bits apply);
 * bit 5: 1 = fp16 prefill perf mode — NOT bit-exact: the layer mat-muls of batches without logits (prefill chunks) run
 * as dense fp16 GEMMs (fp32 accumulation) on dequantized fp16 copies of the matrices made at first use (+2 bytes per weight), the
 * backend's own matrix-core kernel (csrc/perf16.hip; row lengths must be multiples of 64); RoPE, KV append and attention stay the parity kernels on
 * the FP32 cache, single tokens and tree forwards stay entirely on the parity path.
 * bit 7: 1 = the head of a single-token layer as TWO launches (Q / K / V mat-vec with RoPE + KV append, then the one-launch attention) instead of the fused
 * qkv_attn_kernel (k_qkvattn.hip, round 6: four launches per decode layer where the shape is covered).  Same results bit for bit; bit 4 implies it
 * (the fused launch contains the one-launch attention's exchange, its time-out is the same flag and the same fallback).
 * The environment variable PS_HIP_MODE_OR is OR-ed into every mode (A/B runs of unmodified drivers). */
int ps_hip_model_set_mode(ps_hip_model *m, int mode);

#ifdef __cplusplus
}
#endif
#endif /* PS_HIP_H */
