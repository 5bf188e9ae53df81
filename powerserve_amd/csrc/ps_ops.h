// Launchers of the reference-shaped operator kernels (k_ops.hip) and the fused attention kernels
// (k_attn.hip).  Everything enqueues on `st`; pointers are device pointers.
#pragma once
#include "ps_internal.h"

void psl_mul_mat_f32(hipStream_t st, const ps_tensor *dst, const ps_tensor *a, const ps_tensor *b);
void psl_rms_norm(hipStream_t st, const ps_tensor *dst, const ps_tensor *src, const float *w, float eps);
void psl_rope(hipStream_t st, const ps_tensor *dst, const ps_tensor *src, const float *cache, int n_dims, int neox);
void psl_softmax_ext(hipStream_t st, const ps_tensor *dst, const ps_tensor *src, const float *mask, float scale);
void psl_add(hipStream_t st, const ps_tensor *dst, const ps_tensor *a, const ps_tensor *b);
void psl_dup(hipStream_t st, const ps_tensor *dst, const ps_tensor *src);
void psl_silu_hadamard(hipStream_t st, float *out, const float *g, const float *u, int64_t n);
void psl_get_rows(hipStream_t st, const ps_weight *w, const int32_t *tokens_dev, int n, float *out);
void psl_get_mask(hipStream_t st, float *out, int64_t n_kv, int bs, const int32_t *pos_dev, const uint8_t *tree_dev);
void psl_argmax(hipStream_t st, const float *src, int64_t n, int64_t rows, int32_t *out);

// device-resident per-step state of the fused model path (read by kernels so a captured graph replays)
struct ps_step_state {
    int32_t pos0;    // KV position of the first token of this forward
    int32_t bs;      // tokens in this forward
    int32_t n_out;   // decode: number of ids written so far
    int32_t _pad;
};

struct psl_attn_args {
    int n_heads, n_kv_heads, head_size, n_ctx, neox, n_dims;
    const ps_step_state *state;
    const float *rope_table; // [n_ctx][head_size] (cos, sin) pairs, host-built (ggml.c:15344-15358)
    float *q;                // [bs][n_heads*hs]   in: raw q, out: rotated q
    const float *k, *v;      // [bs][kv_dim]
    float *k_cache;          // [n_ctx][kv_dim]
    float *v_cache;          // [kv_dim][n_ctx]
    float *scores;           // [bs][n_heads][n_ctx] scratch
    float *att;              // [bs][n_heads*hs]
    const uint8_t *tree;     // optional [bs][bs] tree mask
    const int32_t *rope_pos; // optional [bs]: RoPE position of each batch column (default: its cache slot pos0 + i)
    const uint8_t *kv_vis;   // optional [n_ctx]: 0 hides a cached slot (KVCacheInterface::mask / unmask)
    float scale;
    ps_act qact;             // optional (qs != null): the V.p kernel of a batch also leaves `att` quantized (Q8_K, + the fragment copies when qf is set) for the O projection
    int64_t qact_K;          // = n_heads * head_size
    int n_kv_host;           // pos0 + bs when the host knows it at enqueue time (eager forwards), 0 inside a captured graph: sizes the score grids
    int bs_host;             // with n_kv_host: the batch size, so that the batch kernels need not wait for the device-resident state (an L2 miss at the head of every launch)
    _Float16 *k16, *v16;     // optional fp16 mirrors of the caches, both [n_ctx][kv_dim] (fp16-KV decode mode: ps_hip_model_set_mode bit 3)
    float *part;             // [n_heads][FL_SPLITS][head_size + 2] partial (o, m, l) of the split-KV decode attention
    unsigned long long *dbg; // timeline buffer of the single-token kernels (ps_hip_debug_timeline keys 40 / 41), or null
    unsigned *sync;          // [2048] words, zeroed once: [31] the one-launch decode attention's spin-timeout flag
    float *xchg;             // attn_decode2: raw scores in flight between the workgroups of a kv head, [n_kv_heads][4][n_ctx rounded up to 32]; null: not used
    unsigned *tick;          // attn_decode2: [64 * kv head] arrival counters, zeroed once (epoch = ticket / workgroups per head)
    int n_kv_lo;             // attn_decode2: a lower bound of pos0 + 1 known to the host at enqueue time (a prefetch HINT only)
    int kv_stream;           // single-token kernels: 1 = the cached K rows / V channels are read with non-temporal loads (the model's whole cache does not fit the memory-side
                             // cache, so nothing of it survives from token to token and it should not displace what does); 0 = plain loads (a small cache is served from there)
};
void psl_rope_append(hipStream_t st, const psl_attn_args &a, int bs);
void psl_attn_scores(hipStream_t st, const psl_attn_args &a, int bs);
void psl_attn_softmax_pv(hipStream_t st, const psl_attn_args &a, int bs);
bool psl_attn_pv_quantizes(const psl_attn_args &a, int bs); // whether psl_attn_softmax_pv(a, bs) would fill a.qact (shape conditions of the fused epilogue)
bool psl_attn_decode_f16(hipStream_t st, const psl_attn_args &a); // single token over the fp16 mirrors: split-KV online soft-max + combine (NOT bit-exact); false: not covered
bool psl_attn_decode2(hipStream_t st, int n_cu, const psl_attn_args &a); // single token: scores + soft-max + V.p in one launch (counter exchange of the scores, V.p on the matrix cores); false: not covered
size_t psl_attn_decode2_xchg_bytes(int n_kv_heads, int n_ctx);
// Q / K / V mat-vec (RMSNorm + quantizer prologue, RoPE + KV append: g.rope set) AND the single-token attention in one launch (k_qkvattn.hip): four launches per decode
// layer instead of five; false: not covered, the caller issues psk_gemv4 + psl_attn_decode2
bool psk_qkv_attn(hipStream_t st, int n_cu, const psk_gemv_args &g, int64_t K, const psl_attn_args &a);
size_t psl_attn_softmax_pv_lds(const psl_attn_args &a); // dynamic LDS bytes (grows with n_ctx)
// two-stage arg-max (64 partials per row).  With state != NULL the final stage also does the greedy-decode
// bookkeeping: token[0] = id, ids[state->n_out++] = id, state->pos0++.
void psl_argmax2(hipStream_t st, const float *src, int64_t n, int64_t rows, int32_t *out, float *part_v, int *part_i,
                 ps_step_state *state, int32_t *token, int32_t *ids);

// host restatement of ggml_rope_cache_init for positions [0, n_pos) -> table[n_pos][ne0]
void ps_rope_table_host(const ps_rope_params *rp, int64_t ne0, const int32_t *pos, int n_pos, float *table);
