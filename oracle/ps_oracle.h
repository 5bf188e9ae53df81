/* oracle/ps_oracle.h — TEST INFRASTRUCTURE ONLY (see ps_oracle.c header).
 *
 * CPU restatement, in plain C, of the arithmetic on PowerServe's ggml decode hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef PS_ORACLE_H
#define PS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml_type enum values of the types on the path (libs/ggml/include/ggml.h:361-398) */
enum pso_type {
    PSO_F32  = 0,
    PSO_F16  = 1,
    PSO_Q4_0 = 2,
    PSO_Q8_0 = 8,
    PSO_Q4_K = 12,
    PSO_Q5_K = 13,
    PSO_Q6_K = 14,
    PSO_Q8_K = 15,
    PSO_I32  = 26,
};

typedef struct {
    int32_t n_dims, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
    int32_t mode; /* 0 = adjacent pairs, 2 = NEOX */
} pso_rope_params;

typedef struct {
    uint32_t dim, hidden_dim, n_layers, n_heads, n_kv_heads, seq_len, vocab_size, kv_dim, head_size;
    float norm_eps;
    pso_rope_params rope;
} pso_llm_config;

/* ---- type helpers */
size_t pso_row_size(int type, int64_t k);
int64_t pso_blck_size(int type);
size_t pso_type_size(int type);
int pso_vec_dot_type(int type);

float pso_fp16_to_fp32(uint16_t h);
uint16_t pso_fp32_to_fp16(float f);

/* ---- activation quantizers (bit-exact contract) */
void pso_quantize_row_q8_0(const float *x, void *y, int64_t k);
void pso_quantize_row_q8_K(const float *x, void *y, int64_t k);
void pso_from_float(int vdt, const float *x, void *y, int64_t k);

/* ---- dequantizers */
void pso_dequantize_row(int type, const void *x, float *y, int64_t k);

/* ---- dot products: weight row (type) x quantized activation row (vec_dot_type(type)) */
float pso_vec_dot(int type, int64_t n, const void *vx, const void *vy);
/* 0 (default): follow the reference built with -ffp-contract=off (oracle/_ref/libps_ref.so); 1: follow its stock build, GCC's default
 * -ffp-contract=fast (oracle/_ref/libps_ref_fast.so) -- see ps_oracle.c */
void pso_set_contract(int on);
int pso_get_contract(void);
float pso_vec_dot_f32(int64_t n, const float *x, const float *y);

/* ---- ops (contiguous layouts unless stated) */
/* y[N,bs] = W[K,N]^T x[K,bs]; W rows of `type`; act_out (may be NULL) gets bs quantized rows */
void pso_mul_mat(int type, const void *w, int64_t K, int64_t N, const float *x, int64_t bs, float *y, void *act_out,
                 int n_threads);
void pso_rms_norm(const float *x, const float *w, float *y, int64_t ne0, int64_t nrows, float eps);
/* src/dst: [ne0=head_size, ne1=n_heads, ne2=npos] contiguous */
void pso_rope(const float *src, float *dst, int64_t ne0, int64_t ne1, int64_t ne2, const int32_t *pos,
              const pso_rope_params *rp);
void pso_rope_cache(int32_t p, int64_t ne0, const pso_rope_params *rp, float *cache /* [ne0] cos,sin pairs */);
/* x,out: [n_kv, bs, n_heads]; mask [n_kv, bs] */
void pso_softmax_ext(const float *x, const float *mask, float *out, int64_t n_kv, int64_t bs, int64_t n_heads,
                     float scale);
void pso_silu_hadamard(const float *gate, const float *up, float *out, int64_t n);
/* out = a + b, b broadcast over rows when nb < na (b has ne0 elements) */
void pso_add(const float *a, const float *b, float *out, int64_t ne0, int64_t nrows, int b_is_row);
void pso_get_embedding(int type, const void *table, int64_t dim, const int32_t *tokens, int n, float *out);

/* ---- whole model (mirrors LlamaModel/Qwen2Model::forward with the ggml backend) */
typedef struct pso_model pso_model;
pso_model *pso_model_create(const pso_llm_config *cfg, int is_qwen2 /* bias + same graph */, int n_threads);
void pso_model_destroy(pso_model *m);
/* name: GGUF tensor name ("token_embd.weight", "blk.3.attn_q.weight", ...). data is NOT copied. */
int pso_model_set_tensor(pso_model *m, const char *name, int type, const void *data, int64_t ne0, int64_t ne1);
size_t pso_model_kv_position(const pso_model *m);
void pso_model_reset(pso_model *m);
void pso_model_rollback(pso_model *m, size_t n); /* rollback_tokens: the last n slots become free again */
int pso_model_forward(pso_model *m, const int32_t *tokens, int n, const int32_t *pos, int lm_head, float *logits_out);
/* token-tree forward: slots [position, position + n), per-column RoPE positions, tree mask [n][n] (NULL: causal), visibility of
 * the cached slots [n_ctx] (NULL: all); advance = 0 leaves the position where it was */
int pso_model_forward_tree(pso_model *m, const int32_t *tokens, int n, const int32_t *rope_pos, const uint8_t *tree,
                           const uint8_t *kv_vis, int lm_head, float *logits_out, int advance);
void pso_model_kv_move(pso_model *m, size_t dst, size_t src);
void pso_model_kv_advance(pso_model *m, size_t n);
int pso_model_generate(pso_model *m, const int32_t *prompt, int n_prompt, int batch_size, int steps,
                       int32_t *out_tokens, float *logits_out, double *t_prefill_s, double *t_decode_s);
/* read-only access to layer L's caches for tests: K [n_ctx][kv_dim], V [kv_dim][n_ctx] */
const float *pso_model_k_cache(const pso_model *m, int L);
const float *pso_model_v_cache(const pso_model *m, int L);

#ifdef __cplusplus
}
#endif
#endif
