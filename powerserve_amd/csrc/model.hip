// Whole-model fast path: what HIPBackend::plan() lowers the reference's per-layer op sequence to
// (LlamaModel::forward, src/model/llama/llama_model.cpp:52-117; NormAttention::build,
// src/model/module/norm_attention.cpp:26-160; FFN::build, src/model/module/ffn.cpp:22-42).
//
// Launch plan per layer (single token, Q4_K, the headline's shape):
//   [matvec[Wq|Wk|Wv] + RMSNorm + quantizer + RoPE + KV append + attention]  (k_qkvattn.hip; elsewhere the mat-vec, then attn_decode2 or scores + soft-max.V.p)
//   -> matvec[Wo] + quantizer + residual -> matvec[Wgate|Wup] + RMSNorm + quantizer + SiLU*up -> matvec[Wdown] + quantizer + residual
// Batches: quantize_act(rmsnorm) -> mat-mul[Wq|Wk|Wv] -> rope_append -> attn_scores -> soft-max -> V.p (+ quantize) -> mat-mul[Wo] + residual
//   -> quantize_act(rmsnorm) -> mat-mul[Wgate|Wup] SiLU*up -> quantize_act -> mat-mul[Wdown] + residual
// A persistent arena replaces the reference's malloc-per-intermediate (src/executor/executor.cpp:23-45);
// the single-token step is captured once into a hipGraph and replayed (all position-dependent values are
// read from a device-resident ps_step_state).
#include "ps_internal.h"
#include <algorithm>
#include "ps_ops.h"

#include <cstdio>
#include <cstring>


namespace {
__global__ void set_state_kernel(ps_step_state *s, int pos0, int bs, int n_out) {
    s->pos0 = pos0; s->bs = bs; s->n_out = n_out;
}
// prefill in super-chunks: one state per reference-sized chunk of the forward (the attention of chunk k sees n_kv = its own end)
__global__ void set_sub_states_kernel(ps_step_state *s, int pos0, int chunk, int n) {
    const int k = threadIdx.x;
    if (k * chunk < n) { s[k].pos0 = pos0 + k * chunk; s[k].bs = min(chunk, n - k * chunk); s[k].n_out = 0; }
}
__global__ void null_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 0; }
__global__ void kv_move_kernel(float *k, float *v, _Float16 *k16, _Float16 *v16, int kvd, int n_ctx, int dst, int src) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= kvd) return;
    k[(int64_t)dst * kvd + d]   = k[(int64_t)src * kvd + d];
    v[(int64_t)d * n_ctx + dst] = v[(int64_t)d * n_ctx + src];
    if (k16) { k16[(int64_t)dst * kvd + d] = k16[(int64_t)src * kvd + d]; v16[(int64_t)dst * kvd + d] = v16[(int64_t)src * kvd + d]; }
}
} // namespace

struct ps_hip_model {
    ps_hip_ctx *ctx = nullptr;
    ps_llm_config cfg{};
    bool qwen2 = false;
    int max_batch = 1;
    const ps_weight *token_embd = nullptr, *output = nullptr;
    const float *output_norm = nullptr;
    std::vector<const float *> attn_norm, ffn_norm, bq, bk, bv;
    std::vector<const ps_weight *> wq, wk, wv, wo, wg, wu, wd;
    // arena
    float *x = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *att = nullptr, *hb = nullptr, *g1 = nullptr, *u1 = nullptr;
    unsigned *attn_sync = nullptr; // [2048] words: [31] the one-launch attention's rendezvous-timeout flag
    unsigned *attn_flag_host = nullptr; // pinned: [31] travels here behind every single-token forward (async copy on the stream)
    bool attn_unchecked = false;        // such a forward was enqueued and its flag has not been looked at yet
    // pinned staging of the lowered single-token path (Model::decode over the op API: one stream synchronisation per token instead of three)
    int32_t *pin_tokens = nullptr;      // [max_batch] the token behind an async upload; pin_busy: not yet known to have been read
    int32_t *pin_argmax = nullptr;      // [max_batch] arg-max ids copied behind a lowered forward; valid for pin_argmax_n columns once pin_argmax_pending is clear
    bool pin_busy = false, pin_argmax_pending = false;
    int pin_argmax_n = 0;
    bool attn1_disabled = false;        // the one-launch attention timed out once on this device: mode bit 4 is sticky (set_mode ORs it back in; bit 6 re-arms)
    float *attn_xchg = nullptr;    // attn_decode2: scores in flight between workgroups (k_attn.hip)
    unsigned *attn_tick = nullptr; // attn_decode2: [64 * kv head] arrival counters
    size_t graph_hint = 0;                   // KV position the captured step was given as its prefetch hint (n_kv_lo)
    int n_kv_hint = 0;             // capture of a single-token forward: the lower bound of n_kv its prefetch hint is baked with
    int n_kv_host = 0;             // pos0 + bs of the forward being enqueued eagerly (0 while a graph is captured / replayed)
    float *scores = nullptr, *logits = nullptr, *rope_table = nullptr;
    void *act_mem = nullptr;
    std::vector<float *> k_cache, v_cache;
    std::vector<_Float16 *> k16, v16; // fp16 mirrors, both [n_ctx][kv_dim]: allocated when mode bit 3 is first set
    float *attn_part = nullptr;       // split-KV partials of the fp16 decode attention
    ps_step_state *state = nullptr;
    int32_t *tokens_dev = nullptr, *argmax_dev = nullptr, *ids_dev = nullptr;
    float *am_v = nullptr;
    int *am_i = nullptr;
    uint8_t *tree_dev = nullptr;
    int32_t *rope_pos_dev = nullptr; // [max_batch] RoPE positions of a tree forward
    uint8_t *kv_vis_dev = nullptr;   // [n_ctx] 1 = visible (KVCacheInterface::mask / unmask)
    std::vector<uint8_t> kv_vis_host;
    size_t n_hidden = 0;
    size_t position = 0;
    int mode = 0;
    psf16 *pf = nullptr;                // fp16 prefill perf mode (mode bit 5): fp16 copies of the layer matrices, fp16 activation scratch
    std::vector<_Float16 *> hq, hk, hv, ho, hg, hu, hd;
    _Float16 *xh = nullptr;
    ps_step_state *state_sub = nullptr; // [64] per-chunk states of a super-chunk prefill (ps_hip_model_prefill)
    int attn_chunk = 0;                 // > 0 while such a forward is enqueued: the attention runs chunk by chunk
    hipGraphExec_t step_graph = nullptr;
    hipGraphExec_t fwd1_graph[2] = {nullptr, nullptr}; // single-token forward without / with lm_head (the lowered op-API path)
    size_t fwd1_hint[2] = {0, 0};
    uint64_t weight_bytes = 0;
    std::vector<void *> owned;
};

static int mode_env_or();
static void drop_graphs(ps_hip_model *m);
static int dmalloc(ps_hip_model *m, void **p, size_t bytes) {
    ps_hip_ctx *c = m->ctx;
    PS_CHECK(c, ps_dev_malloc(p, bytes ? bytes : 16));
    m->owned.push_back(*p);
    return 0;
}

static int mm(ps_hip_model *m, psk_gemv_args &g, ps_act act, int64_t K, int64_t bs);
// from this many columns on a launch quantizes its activation ONCE into `act` and goes to the batched kernels (2-4 columns: 4.8-4.9 ms
// per 8B forward through the batched kernel, 5.4-6.3 ms through the 4-column mat-vec)
static int gemm_min_bs() {
    static const int v = getenv("PS_GEMM8_MIN_BS") ? atoi(getenv("PS_GEMM8_MIN_BS")) : 2;
    return v;
}
// Launches whose matrices are not all of one lane-major type (stock Q4_K_M files: attn_v / ffn_down / output in Q6_K next
// to Q4_K): one launch per matrix, exactly the reference's op sequence (mat-mul, + bias, silu_hadamard).
static int mm_each(ps_hip_model *m, psk_gemv_args &g, ps_act act, int64_t K, int64_t bs) {
    ps_hip_ctx *c = m->ctx;
    float *tmp[2] = {m->g1, m->u1};
    bool quantized = false; // the K-quants of a group share ONE Q8_K quantization of the activation
    auto quantize_once = [&]() {
        if (g.pro && !quantized) psk_quantize_act(c->stream, PS_Q8_K, g.pro == 1 ? 1 : 0, g.pro_x, nullptr, g.pro_norm_w, g.pro_eps, K, bs, act);
        quantized = true;
    };
    bool all5 = g.n_w > 1;
    for (int i = 0; i < g.n_w; i++) all5 = all5 && g.w[i]->dtype == PS_Q5_K;
    if (all5 && bs == 1 && !g.rope) { // Q5_K_M: Q / K / V and gate / up as one launch each (when k_gemvk.hip did not take them)
        quantize_once();
        psk_gemv6_args a5[3];
        for (int i = 0; i < g.n_w; i++) a5[i] = psk_gemv6_args{g.w[i], g.silu_pair ? tmp[i] : g.out[i], g.ldo[i], g.bias[i], (i == 0 && !g.silu_pair) ? g.residual : nullptr};
        if (int rc = psk_gemv5_multi(c->stream, c->n_cu, a5, g.n_w, act, K, bs)) { c->err = "Q5_K mat-vec launch rc=" + std::to_string(rc); return 2; }
        if (g.silu_pair) psl_silu_hadamard(c->stream, g.out[0], tmp[0], tmp[1], (int64_t)g.ldo[0] * bs);
        return 0;
    }
    for (int i = 0; i < g.n_w; i++) {
        psk_gemv_args s{};
        s.n_w = 1; s.w[0] = g.w[i]; s.out[0] = g.silu_pair ? tmp[i] : g.out[i]; s.bias[0] = g.bias[i];
        s.ldo[0] = g.ldo[i]; s.residual = (i == 0 && !g.silu_pair) ? g.residual : nullptr;
        s.pro = g.pro; s.pro_x = g.pro_x; s.pro_norm_w = g.pro_norm_w; s.pro_eps = g.pro_eps;
        s.rope = g.rope; s.rope_wi0 = g.rope_wi0 + i; // (a Q / K / V group split by type: every launch does its part of RoPE + KV append)
        if (g.w[i]->dtype == PS_Q6_K || g.w[i]->dtype == PS_Q5_K) {
            if (bs == 1) { // one matrix of a mixed group (Q4_K_M: a Q6_K V next to Q4_K Q and K), or a run of Q5_K ones (Q5_K_M: Q and K next to a Q6_K V): its own prologue
                int run = 1;
                while (g.w[i]->dtype == PS_Q5_K && !g.silu_pair && i + run < g.n_w && g.w[i + run]->dtype == PS_Q5_K) run++;
                for (int j = 1; j < run; j++) { s.w[j] = g.w[i + j]; s.out[j] = g.out[i + j]; s.bias[j] = g.bias[i + j]; s.ldo[j] = g.ldo[i + j]; }
                s.n_w = run;
                const int rc = psk_gemvk(c->stream, c->n_cu, s, act, K);
                if (rc == 0) { i += run - 1; continue; }
                s.n_w = 1;
                if (rc != -1 || g.rope) { c->err = "Q5_K / Q6_K mat-vec launch rc=" + std::to_string(rc); return 2; }
            }
            quantize_once();
            psk_gemv6_args a6{g.w[i], s.out[0], s.ldo[0], s.bias[0], s.residual};
            if (int rc = psk_gemv6(c->stream, c->n_cu, a6, act, K, bs)) { c->err = "Q5_K / Q6_K mat-vec launch rc=" + std::to_string(rc); return 2; }
        } else {
            // a run of matrices of one lane-major type goes out as ONE launch (Q4_K_M: Q and K next to a Q6_K V)
            int run = 1;
            while (!g.silu_pair && i + run < g.n_w && g.w[i + run]->dtype == g.w[i]->dtype) run++;
            for (int j = 1; j < run; j++) { s.w[j] = g.w[i + j]; s.out[j] = g.out[i + j]; s.bias[j] = g.bias[i + j]; s.ldo[j] = g.ldo[i + j]; }
            s.n_w = run;
            if (mm(m, s, act, K, bs)) return 2;
            i += run - 1;
            quantized = bs >= gemm_min_bs() && g.pro != 0 && ps_hip_vec_dot_type(g.w[i]->dtype) == PS_Q8_K; // (exactly when mm() quantized into `act`: a narrower launch quantizes in its own prologue, into LDS)
        }
    }
    if (g.silu_pair) psl_silu_hadamard(c->stream, g.out[0], tmp[0], tmp[1], (int64_t)g.ldo[0] * bs);
    return 0;
}

static int mm(ps_hip_model *m, psk_gemv_args &g, ps_act act, int64_t K, int64_t bs) {
    ps_hip_ctx *c = m->ctx;
    if (bs == 1 && (g.w[0]->dtype == PS_Q6_K || g.w[0]->dtype == PS_Q5_K)) { // single token: the launch with its fused prologue / epilogue (k_gemvk.hip)
        const int rc = psk_gemvk(c->stream, c->n_cu, g, act, K); // (-1: mixed types, a pair it does not take, ...)
        if (rc == 0) return 0;
        if (rc != -1) { c->err = "Q5_K / Q6_K mat-vec launch rc=" + std::to_string(rc); return 2; }
    }
    bool each = g.w[0]->dtype == PS_Q6_K || g.w[0]->dtype == PS_Q5_K;
    for (int i = 1; i < g.n_w; i++) each = each || g.w[i]->dtype != g.w[0]->dtype;
    if (each) return mm_each(m, g, act, K, bs);
    const int vdt = ps_hip_vec_dot_type(g.w[0]->dtype);
    const int64_t blk = vdt == PS_Q8_0 ? 32 : 256;
    psk_gemv_args gq = g;
    int64_t step = 4;
    if (bs >= gemm_min_bs()) {
        // batches: the activation is quantized ONCE (with its RMSNorm when the launch carries one) and the weights are
        // streamed once per column group of up to 16 instead of once per 4 columns
        if (g.pro) {
            psk_quantize_act(c->stream, vdt, g.pro == 1 ? 1 : 0, g.pro_x, nullptr, g.pro_norm_w, g.pro_eps, K, bs, act);
            gq.pro = 0; gq.pro_x = nullptr;
        }
        if (!gq.pro) {
            const int rc = psk_gemm8(c->stream, c->n_cu, gq, act, K, bs);
            if (rc == 0) return 0;
            if (rc != -1) { c->err = "gemm launch rc=" + std::to_string(rc); return 2; }
        }
        step = psk_gemv_max_cols(g.w[0]->dtype, K);
    }
    for (int64_t c0 = 0; c0 < bs; c0 += step) {
        const int64_t nb = bs - c0 < step ? bs - c0 : step;
        psk_gemv_args gg = gq;
        for (int i = 0; i < g.n_w; i++) gg.out[i] = g.out[i] + c0 * g.ldo[i];
        if (g.residual) gg.residual = g.residual + c0 * g.ldo[0];
        if (gq.pro) gg.pro_x = g.pro_x + c0 * K;
        ps_act ac = act;
        ac.qs += c0 * K; ac.d += c0 * (K / blk); ac.bs16 += c0 * (K / 16);
        if (int rc = psk_gemv(c->stream, c->n_cu, gg, ac, vdt, K, nb)) { c->err = "gemv launch rc=" + std::to_string(rc); return 2; }
    }
    return 0;
}

// fp16 prefill perf mode: the first use dequantizes every layer matrix into an fp16 copy (rows through get_rows, the logits buffer as
// fp32 scratch); the GEMMs are perf16.hip's own kernel
static int ensure_perf16(ps_hip_model *m) {
    if (m->pf) return 0;
    ps_hip_ctx *c = m->ctx;
    const ps_llm_config &f = m->cfg;
    const int64_t dim = f.dim, hid = f.hidden_dim;
    if (dim % 4 || hid % 4) PS_FAIL(c, "fp16 perf mode: dim and hidden_dim must be multiples of 4");
    // the mode's own GEMM (csrc/perf16.hip) takes row lengths and row counts that are multiples of 64: refused HERE, before +2 bytes per weight are allocated
    if (dim % 64 || hid % 64 || f.kv_dim % 64) PS_FAIL(c, "fp16 perf mode: dim, hidden_dim and kv_dim must be multiples of 64 (csrc/perf16.hip's GEMM tiles)");
    psf16 *pf = nullptr;
    if (int rc = psf16_create(c, &pf)) return rc;
    const int64_t kmax = dim > hid ? dim : hid;
    int cap = (int)((int64_t)m->max_batch * f.vocab_size / kmax);
    if (cap > m->max_batch) cap = m->max_batch;
    if (cap < 1) { psf16_destroy(pf); PS_FAIL(c, "fp16 perf mode: no scratch for the dequantizer"); }
    auto copy = [&](const ps_weight *w, std::vector<_Float16 *> &dst) -> int {
        _Float16 *h = nullptr;
        if (dmalloc(m, (void **)&h, (size_t)w->N * w->K * 2)) return 2;
        dst.push_back(h);
        return psf16_dequantize(c, w, m->logits, m->tokens_dev, cap, h);
    };
    // a failed attempt leaves nothing behind: the fp16 copies made so far are freed (and taken off the model's list of owned allocations), so a retry does
    // not allocate them a second time
    auto fail = [&]() {
        for (auto *v : {&m->hq, &m->hk, &m->hv, &m->ho, &m->hg, &m->hu, &m->hd}) {
            for (_Float16 *h : *v) {
                auto it = std::find(m->owned.begin(), m->owned.end(), (void *)h);
                if (it != m->owned.end()) m->owned.erase(it);
                (void)ps_dev_free(h);
            }
            v->clear();
        }
        psf16_destroy(pf);
        return 2;
    };
    for (auto *v : {&m->hq, &m->hk, &m->hv, &m->ho, &m->hg, &m->hu, &m->hd}) v->clear();
    for (uint32_t L = 0; L < f.n_layers; L++)
        if (copy(m->wq[L], m->hq) || copy(m->wk[L], m->hk) || copy(m->wv[L], m->hv) || copy(m->wo[L], m->ho) || copy(m->wg[L], m->hg) ||
            copy(m->wu[L], m->hu) || copy(m->wd[L], m->hd)) return fail();
    if (!m->xh && dmalloc(m, (void **)&m->xh, (size_t)m->max_batch * kmax * 2)) return fail();
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    m->pf = pf;
    return 0;
}

// enqueue one forward over `bs` tokens whose ids are in tokens_dev and whose state is in m->state
int g_kv_stream_force = getenv("PS_KV_STREAM") ? atoi(getenv("PS_KV_STREAM")) : -1; // ps_hip_debug_set(9, v)
static int enqueue_forward(ps_hip_model *m, int bs, bool lm_head, bool use_tree, bool advance = false, bool use_rope_pos = false) {
    ps_hip_ctx *c = m->ctx;
    hipStream_t st = c->stream;
    const ps_llm_config &f = m->cfg;
    const int64_t dim = f.dim, kvd = f.kv_dim, hid = f.hidden_dim;
    const int64_t Kmax = dim > hid ? dim : hid;
    ps_act act = ps_act_carve(m->act_mem, Kmax, m->max_batch);
    // activation rows are laid out with the row strides of the *current* K, so carve per use:
    auto act_for = [&](int64_t K) { return ps_act_carve(m->act_mem, K, m->max_batch); };
    (void)act;

    psl_get_rows(st, m->token_embd, m->tokens_dev, bs, m->x);
    psl_attn_args aa{};
    aa.n_heads = (int)f.n_heads; aa.n_kv_heads = (int)f.n_kv_heads; aa.head_size = (int)f.head_size; aa.n_ctx = (int)f.seq_len;
    aa.neox = (f.rope.mode & 2) ? 1 : 0; aa.n_dims = f.rope.n_dims; aa.state = m->state; aa.rope_table = m->rope_table;
    aa.q = m->q; aa.k = m->k; aa.v = m->v; aa.scores = m->scores; aa.att = m->att;
    aa.tree = use_tree ? m->tree_dev : nullptr;
    aa.rope_pos = use_rope_pos ? m->rope_pos_dev : nullptr;
    aa.kv_vis = m->n_hidden ? m->kv_vis_dev : nullptr;
    aa.scale = 1.0f / sqrtf((float)f.head_size);
    aa.n_kv_host = m->n_kv_host;
    aa.bs_host = m->n_kv_host > 0 ? bs : 0;
    aa.sync = m->attn_sync;
    // fp16 perf mode (NOT bit-exact, mode bit 5): the layer mat-muls of a prefill batch as dense fp16 GEMMs (perf16.hip); single tokens,
    // tree forwards and anything that produces logits stay on the parity kernels
    const bool f16 = (m->mode & 32) && bs >= 2 && !lm_head && !use_tree && !use_rope_pos;
    if (f16) if (int rc = ensure_perf16(m)) return rc;
    const bool one_launch = (m->mode & 16) == 0; // default; mode bit 4 switches attn_decode2 off (two launches)
    aa.xchg = m->attn_xchg; aa.tick = m->attn_tick;
    // Single-token attention: are the cached K rows / V channels read with non-temporal loads (psl_attn_args::kv_stream)?  Measured (profiles/r06_kv_stream_threshold.txt,
    // r06_ad2_nt_small_models.txt): the fused Q / K / V + attention launch is faster with them at every cache length tried (8B: +1.0 % behind a 256-token prompt, +1.6 % behind
    // 1024, +2.2 % behind 2048); the attention launch of its own (attn_decode2) gains where the whole cache -- every layer, K and V -- is more than the memory-side cache keeps
    // from token to token (8B behind 2048 tokens, 537 MB: +0.8 %) and loses where it fits (Llama-3.2-1B behind 512 tokens, 42 MB: -2.2 %).  PS_KV_STREAM=0 / 1 forces both (A/B).
    const int kvs_force = g_kv_stream_force; // (PS_KV_STREAM / ps_hip_debug_set(9, v): -1 by the rule below, 0 / 1 forced)
    const int64_t n_kv_now = m->n_kv_host > 0 ? m->n_kv_host : (m->n_kv_hint > 0 ? m->n_kv_hint : (int64_t)m->position);
    const int kvs_fused = kvs_force >= 0 ? kvs_force : 1;
    const int kvs_own = kvs_force >= 0 ? kvs_force : ((int64_t)f.n_layers * 2 * n_kv_now * kvd * 4 > ((int64_t)160 << 20) ? 1 : 0);
    aa.kv_stream = kvs_own;
    aa.n_kv_lo = m->n_kv_host > 0 ? m->n_kv_host : (m->n_kv_hint > 0 ? m->n_kv_hint : (int)m->position); // (a hint: rows below it are requested before the device-side position has arrived)

    for (uint32_t L = 0; L < f.n_layers; L++) {
        ps_act a1 = act_for(dim);
        psk_gemv_args g{};
        g.n_w = 3;
        g.w[0] = m->wq[L]; g.w[1] = m->wk[L]; g.w[2] = m->wv[L];
        g.out[0] = m->q; g.out[1] = m->k; g.out[2] = m->v;
        g.ldo[0] = dim; g.ldo[1] = kvd; g.ldo[2] = kvd;
        if (m->qwen2) { g.bias[0] = m->bq[L]; g.bias[1] = m->bk[L]; g.bias[2] = m->bv[L]; }
        g.pro = 1; g.pro_x = m->x; g.pro_norm_w = m->attn_norm[L]; g.pro_eps = f.norm_eps; // RMSNorm + quantize in the prologue
        aa.k_cache = m->k_cache[L]; aa.v_cache = m->v_cache[L];
        const bool kv16 = (m->mode & 8) && !m->k16.empty();
        aa.k16 = kv16 ? m->k16[L] : nullptr; aa.v16 = kv16 ? m->v16[L] : nullptr; aa.part = m->attn_part;
        // single token, adjacent-pair RoPE: rotation and the KV append ride in the mat-vec epilogue
        // (mixed types -- Q4_K_M's Q6_K V next to Q4_K Q and K -- and pure Q6_K: one launch per run of a type, each with its part
        //  of the epilogue, psk_gemv_args::rope_wi0)
        auto rope_part_ok = [&](const ps_weight *w) {
            return (w->dtype == PS_Q4_K && psk_gemv4_covers(dim)) || ((w->dtype == PS_Q5_K || w->dtype == PS_Q6_K) && psk_gemvk_covers(w->dtype, dim) && w->N % 8 == 0);
        };
        const bool qkv_same = m->wk[L]->dtype == m->wq[L]->dtype && m->wv[L]->dtype == m->wq[L]->dtype;
        const bool fuse_rope = bs == 1 && !aa.neox && ((qkv_same && psk_gemv_rope_ok(m->wq[L]->dtype, dim)) || (rope_part_ok(m->wq[L]) && rope_part_ok(m->wk[L]) && rope_part_ok(m->wv[L])));
        psk_rope_kv rk{m->state, m->rope_table, aa.k_cache, aa.v_cache, (int)f.head_size, (int)f.rope.n_dims, (int)f.seq_len, (int)kvd, aa.rope_pos, aa.k16, aa.v16};
        // batches of Q4_K weights: the same, in the chunk mat-mul's epilogue (k_gemm4k.hip)
        const bool fuse_rope_b = bs > 1 && !aa.neox && psk_gemm4k_rope_ok(g, dim, bs);
        if ((fuse_rope || fuse_rope_b) && !f16) g.rope = &rk;
        bool fused_qa = false;
        if (f16) {
            psf16_rmsnorm_to_h(st, m->x, m->attn_norm[L], f.norm_eps, dim, bs, m->xh);
            {
                const _Float16 *W3[3] = {m->hq[L], m->hk[L], m->hv[L]};
                float *o3[3] = {m->q, m->k, m->v};
                const int64_t n3[3] = {dim, kvd, kvd};
                if (psf16_gemm_n(c, m->pf, 3, W3, n3, dim, m->xh, bs, o3, n3, 0.f)) return 2; // Q, K, V: one launch
            }
            if (m->qwen2) { psf16_add_bias(st, m->q, m->bq[L], dim, bs); psf16_add_bias(st, m->k, m->bk[L], kvd, bs); psf16_add_bias(st, m->v, m->bv[L], kvd, bs); }
        } else if (bs == 1 && !use_tree && one_launch && !kv16 && g.rope && (m->mode & 128) == 0 &&
                   (aa.dbg = psk_gemv_dbg_buf(10, 3), aa.kv_stream = kvs_fused, psk_qkv_attn(st, c->n_cu, g, dim, aa))) { // timeline key 43
            // the head of the layer in ONE launch: Q / K / V mat-vec + RoPE + KV append + single-token attention (k_qkvattn.hip); mode bit 7 brings the two launches back
            aa.dbg = nullptr; aa.kv_stream = kvs_own;
            fused_qa = true;
        } else if (mm(m, g, a1, dim, bs)) return 2;

        aa.kv_stream = kvs_own; aa.dbg = nullptr; // (also when the fused launch declined the shape)
        if (f16 || (!fuse_rope && !fuse_rope_b)) psl_rope_append(st, aa, bs);
        bool att_quantized = false;
        if (fused_qa) {
        } else if (bs == 1 && !use_tree && kv16 && psl_attn_decode_f16(st, aa)) {
            // fp16-KV decode mode (not bit-exact): split-KV online soft-max over the fp16 mirrors
        } else if (bs == 1 && !use_tree && one_launch && (aa.dbg = psk_gemv_dbg_buf(10, 2), psl_attn_decode2(st, c->n_cu, aa))) { // timeline key 42
            aa.dbg = nullptr;
        } else if (m->attn_chunk > 0 && bs > m->attn_chunk) {
            // a super-chunk of several reference-sized chunks (ps_hip_model_prefill): the mat-muls above and below take all bs columns
            // at once (they are column-wise: chunking cannot change their bits), the attention runs chunk by chunk with the n_kv the
            // reference's chunk would have seen -- its soft-max and V.p sums depend on where a chunk ends
            for (int c0 = 0, k = 0; c0 < bs; c0 += m->attn_chunk, k++) {
                const int nb = bs - c0 < m->attn_chunk ? bs - c0 : m->attn_chunk;
                psl_attn_args as = aa;
                as.state = m->state_sub + k;
                as.q = m->q + (int64_t)c0 * dim; as.att = m->att + (int64_t)c0 * dim;
                as.n_kv_host = m->n_kv_host > 0 ? m->n_kv_host - (bs - c0 - nb) : 0;
                as.bs_host = as.n_kv_host > 0 ? nb : 0;
                as.qact = ps_act{}; as.dbg = nullptr;
                psl_attn_scores(st, as, nb);
                psl_attn_softmax_pv(st, as, nb);
            }
        } else {
            aa.dbg = bs == 1 ? psk_gemv_dbg_buf(10, 0) : nullptr; // timeline key 40
            psl_attn_scores(st, aa, bs);
            aa.dbg = bs == 1 ? psk_gemv_dbg_buf(10, 1) : nullptr; // timeline key 41
            // batches: the V.p kernel can leave `att` quantized for the O projection (one launch less per layer) when the O weight
            // takes Q8_K activations through the batched path
            aa.qact = ps_act{}; aa.qact_K = dim;
            if (!f16 && bs >= 2 && ps_hip_vec_dot_type(m->wo[L]->dtype) == PS_Q8_K && dim % 256 == 0) {
                aa.qact = a1;
                if (!(bs >= ps_gemm4k_min_cols() && dim % 1024 == 0)) { aa.qact.qf = nullptr; aa.qact.mf = nullptr; } // (as psk_quantize_act decides)
                if (!psl_attn_pv_quantizes(aa, bs)) aa.qact = ps_act{};
            }
            att_quantized = aa.qact.qs != nullptr;
            psl_attn_softmax_pv(st, aa, bs);
            aa.qact = ps_act{};
            aa.dbg = nullptr;
        }

        psk_gemv_args go{};
        go.n_w = 1; go.w[0] = m->wo[L]; go.out[0] = m->x; go.ldo[0] = dim; go.residual = m->x;
        go.pro = 2; go.pro_x = m->att;
        if (att_quantized) { go.pro = 0; go.pro_x = nullptr; } // (the V.p kernel left `att` quantized in a1)
        ps_act a2 = act_for(hid);
        const int vdt_d = ps_hip_vec_dot_type(m->wd[L]->dtype);
        psk_gemv_args gf{};
        gf.n_w = 2; gf.w[0] = m->wg[L]; gf.w[1] = m->wu[L]; gf.out[0] = m->hb; gf.out[1] = m->hb; gf.ldo[0] = hid; gf.ldo[1] = hid;
        gf.silu_pair = 1;
        gf.pro = 1; gf.pro_x = m->x; gf.pro_norm_w = m->ffn_norm[L]; gf.pro_eps = f.norm_eps;
        psk_gemv_args gd{};
        gd.n_w = 1; gd.w[0] = m->wd[L]; gd.out[0] = m->x; gd.ldo[0] = dim; gd.residual = m->x;
        if (f16) { // x += O att;  x += down(silu(gate x') * up x'),  x' = rmsnorm(x): residuals through beta = 1
            psf16_to_h(st, m->att, (int64_t)bs * dim, m->xh);
            if (psf16_gemm(c, m->pf, m->ho[L], dim, dim, m->xh, bs, m->x, dim, 1.f)) return 2;
            psf16_rmsnorm_to_h(st, m->x, m->ffn_norm[L], f.norm_eps, dim, bs, m->xh);
            {
                const _Float16 *W2[2] = {m->hg[L], m->hu[L]};
                float *o2[2] = {m->g1, m->u1};
                const int64_t n2[2] = {hid, hid};
                if (psf16_gemm_n(c, m->pf, 2, W2, n2, dim, m->xh, bs, o2, n2, 0.f)) return 2; // gate, up: one launch
            }
            psf16_silu_mul_to_h(st, m->g1, m->u1, (int64_t)bs * hid, m->xh);
            if (psf16_gemm(c, m->pf, m->hd[L], dim, hid, m->xh, bs, m->x, dim, 1.f)) return 2;
            continue;
        }
        if (mm(m, go, a1, dim, bs)) return 2;
        if (mm(m, gf, a1, dim, bs)) return 2;
        if (psk_gemv_lds_col_bytes(m->wd[L]->dtype, hid) * (bs == 1 ? 1 : 4) <= 64 * 1024 && (hid <= 8192 || bs == 1)) {
            gd.pro = 2; gd.pro_x = m->hb; // short rows: quantize in the prologue
        } else {
            psk_quantize_act(st, vdt_d, 0, m->hb, nullptr, nullptr, 0.f, hid, bs, a2);
        }
        if (mm(m, gd, a2, hid, bs)) return 2;
    }
    if (lm_head) {
        const ps_weight *ow = m->output ? m->output : m->token_embd; // tied lm_head (weights.hpp:67-68)
        const int vdt = ps_hip_vec_dot_type(ow->dtype);
        ps_act a1 = act_for(dim);
        (void)vdt;
        psk_gemv_args gl{};
        gl.n_w = 1; gl.w[0] = ow; gl.out[0] = m->logits; gl.ldo[0] = f.vocab_size;
        gl.pro = 1; gl.pro_x = m->x; gl.pro_norm_w = m->output_norm; gl.pro_eps = f.norm_eps;
        if (mm(m, gl, a1, dim, bs)) return 2;
        psl_argmax2(st, m->logits, f.vocab_size, bs, m->argmax_dev, m->am_v, m->am_i, advance ? m->state : nullptr, m->tokens_dev, m->ids_dev);
    }
    PS_CHECK(c, hipGetLastError());
    return 0;
}

extern "C" {

int ps_hip_model_create(ps_hip_ctx *c, const ps_model_desc *d, ps_hip_model **out) {
    *out = nullptr;
    const ps_llm_config &f = d->cfg;
    if (f.n_kv_heads == 0 || f.n_heads == 0 || f.n_layers == 0 || f.seq_len == 0) PS_FAIL(c, "model_create: zero-sized configuration");
    if (f.n_heads % f.n_kv_heads || f.head_size * f.n_heads != f.dim || f.head_size * f.n_kv_heads != f.kv_dim)
        PS_FAIL(c, "model_create: inconsistent head configuration");
    { // the descriptor is trusted by every kernel: weight handles must exist and agree with the configuration
        auto okw = [](const ps_weight *w, int64_t K, int64_t N) {
            return w && w->K == K && w->N == N && w->qs && (w->dtype == PS_Q4_0 || w->dtype == PS_Q8_0 || w->dtype == PS_Q4_K || w->dtype == PS_Q5_K || w->dtype == PS_Q6_K);
        };
        if (!d->token_embd || d->token_embd->K != f.dim || d->token_embd->N != f.vocab_size || !d->token_embd->qs) PS_FAIL(c, "model_create: token_embd missing or of the wrong shape");
        if (d->output && !okw(d->output, f.dim, f.vocab_size)) PS_FAIL(c, "model_create: output.weight of the wrong shape / type");
        if (!d->output && !okw(d->token_embd, f.dim, f.vocab_size)) PS_FAIL(c, "model_create: tied lm_head needs a quantized token_embd");
        if (!d->output_norm || !d->attn_norm || !d->ffn_norm || !d->attn_q || !d->attn_k || !d->attn_v || !d->attn_output || !d->ffn_gate || !d->ffn_up || !d->ffn_down)
            PS_FAIL(c, "model_create: null weight table");
        for (uint32_t i = 0; i < f.n_layers; i++) {
            if (!d->attn_norm[i] || !d->ffn_norm[i]) PS_FAIL(c, "model_create: null norm weights in layer " + std::to_string(i));
            if (!okw(d->attn_q[i], f.dim, f.dim) || !okw(d->attn_k[i], f.dim, f.kv_dim) || !okw(d->attn_v[i], f.dim, f.kv_dim) || !okw(d->attn_output[i], f.dim, f.dim) ||
                !okw(d->ffn_gate[i], f.dim, f.hidden_dim) || !okw(d->ffn_up[i], f.dim, f.hidden_dim) || !okw(d->ffn_down[i], f.hidden_dim, f.dim))
                PS_FAIL(c, "model_create: weight of layer " + std::to_string(i) + " missing or of the wrong shape / type (model.json and weights.gguf disagree?)");
            if (d->is_qwen2 && (!d->attn_q_bias || !d->attn_k_bias || !d->attn_v_bias || !d->attn_q_bias[i] || !d->attn_k_bias[i] || !d->attn_v_bias[i]))
                PS_FAIL(c, "model_create: qwen2 needs the attention biases");
        }
    }
    if (f.head_size % 32 || f.head_size > 128) PS_FAIL(c, "model_create: head_size must be 32, 64, 96 or 128");
    if (f.n_heads / f.n_kv_heads > 8) PS_FAIL(c, "model_create: GQA ratio > 8 not supported");
    if ((int)f.rope.n_dims != (int)f.head_size) PS_FAIL(c, "model_create: rope n_dims != head_size (reference asserts the same, norm_attention.cpp:38)");
    if (f.seq_len % 4) PS_FAIL(c, "model_create: n_ctx must be a multiple of 4");
    if ((size_t)(f.n_heads / f.n_kv_heads) * f.seq_len * 4 + 44 * 1024 > 158 * 1024) PS_FAIL(c, "model_create: n_ctx too large for the LDS-resident softmax rows (cap n_ctx)");
    if (f.n_heads / f.n_kv_heads > 8) PS_FAIL(c, "model_create: more than 8 query heads per kv head");
    PS_CHECK(c, hipSetDevice(c->device));
    auto m = new ps_hip_model();
    m->ctx = c; m->cfg = f; m->qwen2 = d->is_qwen2 != 0; m->max_batch = d->max_batch > 0 ? d->max_batch : 1;
    m->mode = mode_env_or() & ~8; // (the fp16-KV mode allocates: only through set_mode)
    m->token_embd = d->token_embd; m->output = d->output; m->output_norm = d->output_norm;
    const uint32_t L = f.n_layers;
    auto cpf = [&](std::vector<const float *> &v, const float *const *src) { v.assign(L, nullptr); if (src) for (uint32_t i = 0; i < L; i++) v[i] = src[i]; };
    auto cpw = [&](std::vector<const ps_weight *> &v, const ps_weight *const *src) { v.assign(L, nullptr); for (uint32_t i = 0; i < L; i++) v[i] = src[i]; };
    cpf(m->attn_norm, d->attn_norm); cpf(m->ffn_norm, d->ffn_norm);
    cpf(m->bq, d->attn_q_bias); cpf(m->bk, d->attn_k_bias); cpf(m->bv, d->attn_v_bias);
    cpw(m->wq, d->attn_q); cpw(m->wk, d->attn_k); cpw(m->wv, d->attn_v); cpw(m->wo, d->attn_output);
    cpw(m->wg, d->ffn_gate); cpw(m->wu, d->ffn_up); cpw(m->wd, d->ffn_down);
    m->weight_bytes = (m->output ? m->output : m->token_embd)->gguf_bytes;
    for (uint32_t i = 0; i < L; i++)
        m->weight_bytes += m->wq[i]->gguf_bytes + m->wk[i]->gguf_bytes + m->wv[i]->gguf_bytes + m->wo[i]->gguf_bytes +
                           m->wg[i]->gguf_bytes + m->wu[i]->gguf_bytes + m->wd[i]->gguf_bytes;

    const size_t mb = (size_t)m->max_batch, dim = f.dim, kvd = f.kv_dim, hid = f.hidden_dim, nctx = f.seq_len;
    auto fail = [&]() { ps_hip_model_destroy(m); return 1; };
    if (dmalloc(m, (void **)&m->x, mb * dim * 4) || dmalloc(m, (void **)&m->q, mb * dim * 4) ||
        dmalloc(m, (void **)&m->k, mb * kvd * 4) || dmalloc(m, (void **)&m->v, mb * kvd * 4) ||
        dmalloc(m, (void **)&m->att, mb * dim * 4) || dmalloc(m, (void **)&m->hb, mb * hid * 4) ||
        dmalloc(m, (void **)&m->g1, mb * hid * 4) || dmalloc(m, (void **)&m->u1, mb * hid * 4) ||
        dmalloc(m, (void **)&m->attn_sync, 2048 * 4) ||
        dmalloc(m, (void **)&m->attn_xchg, psl_attn_decode2_xchg_bytes((int)f.n_kv_heads, (int)nctx)) || dmalloc(m, (void **)&m->attn_tick, (size_t)f.n_kv_heads * 64 * 4) ||
        dmalloc(m, (void **)&m->scores, mb * f.n_heads * nctx * 4) || dmalloc(m, (void **)&m->logits, mb * f.vocab_size * 4) ||
        dmalloc(m, (void **)&m->rope_table, nctx * f.head_size * 4) ||
        dmalloc(m, &m->act_mem, ps_act_bytes(dim > hid ? dim : hid, mb)) ||
        dmalloc(m, (void **)&m->state, sizeof(ps_step_state)) || dmalloc(m, (void **)&m->state_sub, 64 * sizeof(ps_step_state)) || dmalloc(m, (void **)&m->tokens_dev, mb * 4) ||
        dmalloc(m, (void **)&m->argmax_dev, mb * 4) || dmalloc(m, (void **)&m->ids_dev, (nctx + 1) * 4) ||
        dmalloc(m, (void **)&m->tree_dev, mb * mb) || dmalloc(m, (void **)&m->rope_pos_dev, mb * 4) || dmalloc(m, (void **)&m->kv_vis_dev, nctx) || dmalloc(m, (void **)&m->am_v, mb * 64 * 4) || dmalloc(m, (void **)&m->am_i, mb * 64 * 4))
        return fail();
    (void)hipMemsetAsync(m->attn_sync, 0, 2048 * 4, c->stream);
    if (hipHostMalloc((void **)&m->attn_flag_host, 64, hipHostMallocDefault) != hipSuccess) return fail();
    *m->attn_flag_host = 0;
    if (hipHostMalloc((void **)&m->pin_tokens, (2 * mb + 16) * 4, hipHostMallocDefault) != hipSuccess) return fail();
    m->pin_argmax = m->pin_tokens + mb;
    (void)hipMemsetAsync(m->attn_tick, 0, (size_t)f.n_kv_heads * 64 * 4, c->stream);
    (void)hipMemsetAsync(m->kv_vis_dev, 1, nctx, c->stream);
    m->kv_vis_host.assign(nctx, 1);
    m->k_cache.assign(L, nullptr); m->v_cache.assign(L, nullptr);
    for (uint32_t i = 0; i < L; i++) {
        if (dmalloc(m, (void **)&m->k_cache[i], nctx * kvd * 4) || dmalloc(m, (void **)&m->v_cache[i], nctx * kvd * 4)) return fail();
        (void)hipMemsetAsync(m->k_cache[i], 0, nctx * kvd * 4, c->stream);
        (void)hipMemsetAsync(m->v_cache[i], 0, nctx * kvd * 4, c->stream);
    }
    // RoPE table for every cache position, host-built with the reference recurrence
    std::vector<float> tab(nctx * f.head_size);
    ps_rope_table_host(&f.rope, f.head_size, nullptr, (int)nctx, tab.data());
    if (hipMemcpyAsync(m->rope_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail();
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail();
    *out = m;
    return 0;
}

void ps_hip_model_destroy(ps_hip_model *m) {
    if (!m) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    drop_graphs(m);
    if (m->attn_flag_host) (void)hipHostFree(m->attn_flag_host);
    if (m->pin_tokens) (void)hipHostFree(m->pin_tokens);
    psf16_destroy(m->pf);
    for (void *p : m->owned) (void)ps_dev_free(p);
    delete m;
}

size_t ps_hip_model_kv_position(const ps_hip_model *m) { return m->position; }
int ps_hip_model_max_batch(const ps_hip_model *m) { return m->max_batch; }
static void unmask_range(ps_hip_model *m, size_t from, size_t n);
extern "C" int ps_hip_model_sync_check(ps_hip_model *m);
static int settle_pending(ps_hip_model *m);
// slots at or behind the position are never consulted through the visibility table (the causal / tree mask governs them)
// and KVCache::advance_tokens / append un-hides what it walks over (core/kv_cache.hpp:249-255): a rollback or truncate
// leaves no hidden slot behind the new position, so a model that served as a speculative draft can decode again
int ps_hip_model_kv_truncate(ps_hip_model *m, size_t n) {
    if (int rc = settle_pending(m)) return rc; // (every KV entry point: an unconsumed lowered forward's time-out belongs to ITS caller, not to the next forward)
    if (n < m->position) m->position = n;
    unmask_range(m, m->position, m->cfg.seq_len - m->position);
    return 0;
}
// KVCache::advance_tokens unmasks the slots it walks over (core/kv_cache.hpp:249-255)
static void unmask_range(ps_hip_model *m, size_t from, size_t n) {
    if (!m->n_hidden) return;
    for (size_t i = from; i < from + n && i < m->cfg.seq_len; i++) {
        if (!m->kv_vis_host[i]) {
            m->kv_vis_host[i] = 1;
            m->n_hidden--;
            (void)hipMemcpyAsync(m->kv_vis_dev + i, &m->kv_vis_host[i], 1, hipMemcpyHostToDevice, m->ctx->stream);
        }
    }
}
int ps_hip_model_kv_advance(ps_hip_model *m, size_t n) {
    if (int rc = ps_hip_model_sync_check(m)) return rc; // (an unsynchronised lowered forward: its rows must be valid before they count)
    if (m->position + n > m->cfg.seq_len) { m->ctx->err = "kv_advance: KV cache is full (n_ctx)"; return 2; }
    unmask_range(m, m->position, n);
    m->position += n;
    return 0;
}
int ps_hip_model_kv_rollback(ps_hip_model *m, size_t n) {
    if (int rc = settle_pending(m)) return rc;
    if (n > m->position) { m->ctx->err = "kv_rollback: more tokens than cached"; return 2; }
    m->position -= n;
    unmask_range(m, m->position, m->cfg.seq_len - m->position);
    return 0;
}
int ps_hip_model_kv_move(ps_hip_model *m, size_t dst, size_t src) {
    if (int rc = settle_pending(m)) return rc; // (rows of a forward whose result is invalid must not be copied)
    if (dst == src) return 0;
    if (dst >= m->cfg.seq_len || src >= m->cfg.seq_len) { m->ctx->err = "kv_move: index out of range"; return 2; }
    for (uint32_t L = 0; L < m->cfg.n_layers; L++)
        hipLaunchKernelGGL(kv_move_kernel, dim3((m->cfg.kv_dim + 255) / 256), dim3(256), 0, m->ctx->stream, m->k_cache[L],
                           m->v_cache[L], m->k16.empty() ? nullptr : m->k16[L], m->v16.empty() ? nullptr : m->v16[L], (int)m->cfg.kv_dim, (int)m->cfg.seq_len, (int)dst, (int)src);
    return 0;
}
// ---- the rest of KVCacheInterface (core/kv_cache.hpp:120-162; see include/ps_hip.h for why they reduce to these)
int ps_hip_model_kv_copy(ps_hip_model *m, size_t dst_cache_index, size_t src_token_index) {
    return ps_hip_model_kv_move(m, dst_cache_index, m->position + src_token_index);
}
int ps_hip_model_kv_save_tokens(ps_hip_model *m, size_t n) {
    if (int rc = settle_pending(m)) return rc;
    if (m->position + n > m->cfg.seq_len) { m->ctx->err = "kv_save_tokens: the length of kvcache is up to the preset threshold (n_ctx)"; return 2; }
    return 0;
}
int ps_hip_model_kv_unmask_tokens(ps_hip_model *m, size_t n) {
    if (int rc = settle_pending(m)) return rc;
    if (m->position + n > m->cfg.seq_len) { m->ctx->err = "kv_unmask_tokens: the length of kvcache is up to the preset threshold (n_ctx)"; return 2; }
    unmask_range(m, m->position, n);
    return 0;
}
int ps_hip_model_kv_append_tokens(ps_hip_model *m, size_t n, size_t *old_position) {
    const size_t old = m->position;
    if (int rc = ps_hip_model_kv_save_tokens(m, n)) return rc;
    if (int rc = ps_hip_model_kv_unmask_tokens(m, n)) return rc;
    if (int rc = ps_hip_model_kv_advance(m, n)) return rc;
    if (old_position) *old_position = old;
    return 0;
}

static int model_forward_impl(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *pos, const uint8_t *tree, int lm_head,
                              int32_t *argmax_host, bool advance, bool retried = false);
// The one-launch attentions wait for each other's workgroups inside the launch (bounded); a wait that gave up raises
// attn_sync[31].  Read it behind every synchronised single-token forward, clear it, and fall back to the two launches
// from here on (mode bit 4): the forward that timed out has no valid result.  Expects the stream to be idle.
static void drop_graphs(ps_hip_model *m) { // every captured launch plan (they bake the mode, the prefetch hint and the visibility pointers in)
    if (m->step_graph) { (void)hipGraphExecDestroy(m->step_graph); m->step_graph = nullptr; }
    for (auto &g : m->fwd1_graph) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
}
// Every single-token forward that may have used the one-launch attention is followed, on the stream, by a 4-byte copy of the flag into pinned
// host memory (note_single_token): looking at it costs no round trip once the stream is idle.
int g_force_attn_timeout = 0; // ps_hip_debug_set(5, n): the next n single-token forwards on the one-launch attention report a time-out (tests of the retry paths)
static void note_single_token(ps_hip_model *m) {
    if (m->mode & 16) return; // the one-launch form is not in use
    if (g_force_attn_timeout > 0) { g_force_attn_timeout--; (void)hipMemsetAsync(m->attn_sync + 31, 1, 1, m->ctx->stream); } // (word = 1, behind the forward's launches)
    (void)hipMemcpyAsync(m->attn_flag_host, m->attn_sync + 31, 4, hipMemcpyDeviceToHost, m->ctx->stream);
    m->attn_unchecked = true;
}
// 0: nothing pending or the flag is clear.  PS_HIP_ATTN_TIMEOUT (3): the forward(s) since the last check have no valid result; the model has
// switched to the two-launch attention (mode bit 4, sticky; graphs dropped) and c->err says so -- the callers that still know their inputs run
// them again, ONCE (behind the switch no forward can time out: note_single_token arms nothing under mode bit 4).  1: a HIP error (c->err),
// never retried.  The switch happens before anything that can fail, so a failing memset cannot leave the one-launch form armed with its flag up.
static int check_attn_timeout(ps_hip_model *m, const char *who) {
    ps_hip_ctx *c = m->ctx;
    if (!m->attn_unchecked) return 0;
    m->attn_unchecked = false;
    if (!*(volatile unsigned *)m->attn_flag_host) return 0;
    *(volatile unsigned *)m->attn_flag_host = 0;
    m->mode |= 16;
    m->attn1_disabled = true;
    drop_graphs(m);
    PS_CHECK(c, hipMemset(m->attn_sync + 31, 0, 4));
    c->err = std::string(who) + ": the one-launch attention timed out at its score exchange (GPU shared or partitioned?); this forward has no valid "
             "result, the cache position is unchanged, and the model now uses the two-launch attention (mode bit 4)";
    return PS_HIP_ATTN_TIMEOUT;
}
// A lowered forward that nobody has consumed yet (no kv_advance / sync_check since) must not leave its flag to the next, unrelated forward:
// every entry point settles it first and hands a time-out back to the caller, who still owns that forward's inputs.
static int settle_pending(ps_hip_model *m) {
    if (!m->attn_unchecked) return 0;
    return ps_hip_model_sync_check(m);
}
// The lowered op-API path (ps_hip_model_forward_lowered) returns before its launches have run: whoever consumes its result next -- the
// cache advance (LlamaModel::forward advances right behind Executor::run, llama_model.cpp:109) or a logits read -- looks at the flag.
int ps_hip_model_sync_check(ps_hip_model *m) {
    ps_hip_ctx *c = m->ctx;
    if (!m->attn_unchecked) return 0;
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    m->pin_busy = false; m->pin_argmax_pending = false; // (everything enqueued before this point has run)
    return check_attn_timeout(m, "lowered forward");
}
int ps_hip_model_forward(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *pos, const uint8_t *tree, int lm_head,
                         int32_t *argmax_host) {
    return model_forward_impl(m, tokens, n, pos, tree, lm_head, argmax_host, true);
}
int ps_hip_model_forward_lowered(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *pos, const uint8_t *tree, int lm_head) {
    return model_forward_impl(m, tokens, n, pos, tree, lm_head, nullptr, false);
}
static int model_forward_impl(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *pos, const uint8_t *tree, int lm_head,
                              int32_t *argmax_host, bool advance, bool retried) {
    ps_hip_ctx *c = m->ctx;
    if (int rc = settle_pending(m)) return rc;
    if (n <= 0 || n > m->max_batch) PS_FAIL(c, "model_forward: batch size out of range");
    for (int i = 1; i < n; i++)
        if (pos[i] != pos[0] + i) PS_FAIL(c, "model_forward: positions must be consecutive (KV append is one contiguous copy, norm_attention.cpp:82-91)");
    if (pos[0] < 0 || (size_t)pos[0] + (size_t)n > m->cfg.seq_len) PS_FAIL(c, "model_forward: KV cache is full (n_ctx)");
    for (int i = 0; i < n; i++)
        if (tokens[i] < 0 || (uint32_t)tokens[i] >= m->cfg.vocab_size) PS_FAIL(c, "model_forward: token id out of range");
    PS_CHECK(c, hipSetDevice(c->device));
    m->pin_argmax_n = 0; // (the arg-max ids on the device are about to change)
    if (n == 1 && !tree) { // a decode step: the token travels from a pinned word, nothing to wait for before the launches go out
        if (m->pin_busy) { PS_CHECK(c, hipStreamSynchronize(c->stream)); m->pin_busy = false; m->pin_argmax_pending = false; } // (an upload from that word may still be queued)
        m->pin_tokens[0] = tokens[0];
        PS_CHECK(c, hipMemcpyAsync(m->tokens_dev, m->pin_tokens, 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, c->stream, m->state, pos[0], n, 0);
        m->pin_busy = true;
    } else {
        PS_CHECK(c, hipMemcpyAsync(m->tokens_dev, tokens, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        if (tree) PS_CHECK(c, hipMemcpyAsync(m->tree_dev, tree, (size_t)n * n, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, c->stream, m->state, pos[0], n, 0);
        PS_CHECK(c, hipStreamSynchronize(c->stream)); // tokens/tree may be host temporaries
        m->pin_busy = false; m->pin_argmax_pending = false;
    }
    // A single token through this entry (ModelTokenIterator::decode over the op API, HIPBackend::plan's lowered graph) replays
    // a captured launch plan like ps_hip_model_decode_greedy does: the first call at a position range runs eagerly and captures.
    const int gk = lm_head ? 1 : 0;
    const bool graphable = n == 1 && !tree && (m->mode & 1) == 0 && m->n_hidden == 0;
    if (graphable && m->fwd1_graph[gk] && ((size_t)pos[0] < m->fwd1_hint[gk] || (size_t)pos[0] >= m->fwd1_hint[gk] + 1024)) {
        (void)hipGraphExecDestroy(m->fwd1_graph[gk]); m->fwd1_graph[gk] = nullptr;
    }
    if (graphable && m->fwd1_graph[gk]) {
        PS_CHECK(c, hipGraphLaunch(m->fwd1_graph[gk], c->stream));
    } else {
        m->n_kv_host = pos[0] + n;
        const int rc_fw = enqueue_forward(m, n, lm_head != 0, tree != nullptr);
        if (rc_fw) { m->n_kv_host = 0; return rc_fw; }
        if (graphable) { // capture the same launches (nothing executes); the hint stays a lower bound of n_kv for 1024 positions
            hipGraph_t g = nullptr;
            PS_CHECK(c, hipStreamSynchronize(c->stream));
            m->n_kv_host = 0; m->n_kv_hint = pos[0]; // (no host-side n_kv inside a graph: the score grids cover n_ctx)
            PS_CHECK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_forward(m, 1, lm_head != 0, false);
            const hipError_t e = hipStreamEndCapture(c->stream, &g);
            m->n_kv_hint = 0;
            if (!rc && e == hipSuccess && hipGraphInstantiate(&m->fwd1_graph[gk], g, nullptr, nullptr, 0) == hipSuccess) m->fwd1_hint[gk] = (size_t)pos[0];
            else m->fwd1_graph[gk] = nullptr; // (stay eager)
            if (g) (void)hipGraphDestroy(g);
        }
        m->n_kv_host = 0;
    }
    if (n == 1 && !tree) note_single_token(m);
    if (lm_head && argmax_host) PS_CHECK(c, hipMemcpyAsync(argmax_host, m->argmax_dev, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (lm_head && !advance) { // lowered graph: the ids ride to pinned memory behind the launches; ps_hip_model_argmax finds them there after the cache advance's wait
        PS_CHECK(c, hipMemcpyAsync(m->pin_argmax, m->argmax_dev, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        m->pin_argmax_n = n; m->pin_argmax_pending = true;
    }
    if (!advance) return 0; // lowered graph: ps_hip_model_kv_advance / ps_hip_model_sync_check look at the time-out flag before the result counts
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    m->pin_busy = false; m->pin_argmax_pending = false;
    if (const int rc = check_attn_timeout(m, "model_forward")) { // a time-out: the model is on the two-launch attention now, the same forward once more (inputs are the caller's)
        if (rc != PS_HIP_ATTN_TIMEOUT || retried) return rc;
        c->err.clear();
        return model_forward_impl(m, tokens, n, pos, tree, lm_head, argmax_host, advance, true);
    }
    unmask_range(m, (size_t)pos[0], (size_t)n);
    m->position = (size_t)pos[0] + (size_t)n; // m_kv->advance (llama_model.cpp:109)
    return 0;
}

// ModelTokenIterator's prefill (src/model/model.hpp:147-163): forward(tokens[done .. done + chunk), lm_head = false) chunk after chunk, the
// cache advanced after each -- with the SAME bits, but up to max_batch / chunk reference chunks per launch sequence ("super-chunk"): every
// mat-mul of a layer sees all their columns at once (per-chunk fixed costs paid a quarter as often at chunk 128 / max_batch 512), only the
// attention is evaluated per reference chunk (ps_hip_model::attn_chunk).  8B, chunk 128: 14.4 k -> ~17 k tok/s.
int ps_hip_model_prefill(ps_hip_model *m, const int32_t *tokens, int n, int chunk) {
    ps_hip_ctx *c = m->ctx;
    if (n <= 0) return 0;
    if (int rc = settle_pending(m)) return rc;
    if (chunk <= 0 || chunk > m->max_batch) PS_FAIL(c, "model_prefill: chunk size out of range");
    if (m->position + (size_t)n > m->cfg.seq_len) PS_FAIL(c, "model_prefill: KV cache is full (n_ctx)");
    for (int i = 0; i < n; i++)
        if (tokens[i] < 0 || (uint32_t)tokens[i] >= m->cfg.vocab_size) PS_FAIL(c, "model_prefill: token id out of range");
    PS_CHECK(c, hipSetDevice(c->device));
    m->pin_argmax_n = 0;
    int per = m->max_batch / chunk; // reference chunks per super-chunk
    if (per > 64) per = 64;
    if (per < 1) per = 1;
    bool retried = false;
    for (int done = 0; done < n;) {
        const int ns = n - done < per * chunk ? n - done : per * chunk;
        PS_CHECK(c, hipMemcpyAsync(m->tokens_dev, tokens + done, (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, c->stream, m->state, (int)m->position, ns, 0);
        hipLaunchKernelGGL(set_sub_states_kernel, dim3(1), dim3(64), 0, c->stream, m->state_sub, (int)m->position, chunk, ns);
        m->n_kv_host = (int)m->position + ns;
        m->attn_chunk = chunk;
        const int rc = enqueue_forward(m, ns, false, false);
        m->attn_chunk = 0; m->n_kv_host = 0;
        if (rc) return rc;
        if (ns == 1) note_single_token(m);
        PS_CHECK(c, hipStreamSynchronize(c->stream));
        if (const int rc = check_attn_timeout(m, "model_prefill")) { // a one-token tail: once more, with the two launches
            if (rc != PS_HIP_ATTN_TIMEOUT || retried) return rc;
            retried = true; c->err.clear(); continue;
        }
        unmask_range(m, m->position, (size_t)ns);
        m->position += (size_t)ns;
        done += ns;
    }
    return 0;
}

static int forward_tree_impl(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *rope_pos, const uint8_t *tree, int lm_head,
                             int32_t *argmax_host, int advance, bool retried);
int ps_hip_model_forward_tree(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *rope_pos, const uint8_t *tree, int lm_head,
                              int32_t *argmax_host, int advance) {
    return forward_tree_impl(m, tokens, n, rope_pos, tree, lm_head, argmax_host, advance, false);
}
static int forward_tree_impl(ps_hip_model *m, const int32_t *tokens, int n, const int32_t *rope_pos, const uint8_t *tree, int lm_head,
                             int32_t *argmax_host, int advance, bool retried) {
    ps_hip_ctx *c = m->ctx;
    if (int rc = settle_pending(m)) return rc;
    if (n <= 0 || n > m->max_batch) PS_FAIL(c, "model_forward_tree: batch size out of range");
    if (m->position + (size_t)n > m->cfg.seq_len) PS_FAIL(c, "model_forward_tree: KV cache is full (n_ctx)");
    for (int i = 0; i < n; i++) {
        if (tokens[i] < 0 || (uint32_t)tokens[i] >= m->cfg.vocab_size) PS_FAIL(c, "model_forward_tree: token id out of range");
        if (rope_pos[i] < 0 || (uint32_t)rope_pos[i] >= m->cfg.seq_len) PS_FAIL(c, "model_forward_tree: position out of range");
    }
    PS_CHECK(c, hipSetDevice(c->device));
    m->pin_argmax_n = 0;
    PS_CHECK(c, hipMemcpyAsync(m->tokens_dev, tokens, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    PS_CHECK(c, hipMemcpyAsync(m->rope_pos_dev, rope_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (tree) PS_CHECK(c, hipMemcpyAsync(m->tree_dev, tree, (size_t)n * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, c->stream, m->state, (int)m->position, n, 0);
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    m->n_kv_host = (int)m->position + n;
    const int rc_fw = enqueue_forward(m, n, lm_head != 0, tree != nullptr, false, true);
    m->n_kv_host = 0;
    if (rc_fw) return rc_fw;
    if (n == 1 && !tree) note_single_token(m);
    if (lm_head && argmax_host) PS_CHECK(c, hipMemcpyAsync(argmax_host, m->argmax_dev, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    if (const int rc = check_attn_timeout(m, "model_forward_tree")) {
        if (rc != PS_HIP_ATTN_TIMEOUT || retried) return rc;
        c->err.clear();
        return forward_tree_impl(m, tokens, n, rope_pos, tree, lm_head, argmax_host, advance, true);
    }
    if (advance) { unmask_range(m, m->position, (size_t)n); m->position += (size_t)n; }
    return 0;
}

int ps_hip_model_kv_mask(ps_hip_model *m, size_t index, int visible) {
    ps_hip_ctx *c = m->ctx;
    if (int rc = settle_pending(m)) return rc;
    if (index >= m->cfg.seq_len) PS_FAIL(c, "kv_mask: index out of range");
    const uint8_t v = visible ? 1 : 0;
    if (m->kv_vis_host[index] == v) return 0;
    m->kv_vis_host[index] = v;
    if (v) m->n_hidden--; else m->n_hidden++;
    PS_CHECK(c, hipMemcpyAsync(m->kv_vis_dev + index, &m->kv_vis_host[index], 1, hipMemcpyHostToDevice, c->stream));
    return 0;
}

static int decode_greedy_impl(ps_hip_model *m, int32_t token, int steps, int32_t *out_ids, bool retried);
int ps_hip_model_decode_greedy(ps_hip_model *m, int32_t token, int steps, int32_t *out_ids) { return decode_greedy_impl(m, token, steps, out_ids, false); }
static int decode_greedy_impl(ps_hip_model *m, int32_t token, int steps, int32_t *out_ids, bool retried) {
    ps_hip_ctx *c = m->ctx;
    if (steps <= 0) return 0;
    if (int rc = settle_pending(m)) return rc;
    if (m->n_hidden) PS_FAIL(c, "decode_greedy: hidden KV slots (kv_mask) are not part of the captured step; unmask first");
    if (m->position + (size_t)steps > m->cfg.seq_len) PS_FAIL(c, "decode_greedy: KV cache would overflow n_ctx");
    if (token < 0 || (uint32_t)token >= m->cfg.vocab_size) PS_FAIL(c, "decode_greedy: token id out of range");
    PS_CHECK(c, hipSetDevice(c->device));
    m->pin_argmax_n = 0;
    PS_CHECK(c, hipMemcpyAsync(m->tokens_dev, &token, 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, c->stream, m->state, (int)m->position, 1, 0);
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    int s = 0;
    // the captured step carries the position it was captured at as a prefetch hint (n_kv_lo): keep it a LOWER bound that
    // is not far behind (a stale hint costs time, never results)
    if (m->step_graph && (m->position < m->graph_hint || m->position >= m->graph_hint + 1024)) { (void)hipGraphExecDestroy(m->step_graph); m->step_graph = nullptr; }
    if ((m->mode & 1) == 0 && !m->step_graph) {
        // first step runs eagerly (also performs every one-time hipFuncSetAttribute), then the identical
        // launch sequence is captured; capture itself executes nothing
        if (int rc = enqueue_forward(m, 1, true, false, true)) return rc;
        PS_CHECK(c, hipStreamSynchronize(c->stream));
        s = 1;
        hipGraph_t g = nullptr;
        m->graph_hint = m->position;
        PS_CHECK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_forward(m, 1, true, false, true);
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc || e != hipSuccess) { c->err = "decode_greedy: graph capture failed: " + c->err; if (g) (void)hipGraphDestroy(g); return 2; }
        PS_CHECK(c, hipGraphInstantiate(&m->step_graph, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
    }
    for (; s < steps; s++) {
        if ((m->mode & 1) == 0) {
            PS_CHECK(c, hipGraphLaunch(m->step_graph, c->stream));
        } else {
            if (int rc = enqueue_forward(m, 1, true, false, true)) return rc;
        }
    }
    note_single_token(m);
    PS_CHECK(c, hipMemcpyAsync(out_ids, m->ids_dev, (size_t)steps * 4, hipMemcpyDeviceToHost, c->stream));
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    // a time-out anywhere in the run: the position has not moved and the first token is the caller's -- the whole run once more on the
    // two-launch attention (every cache row it wrote is written again with the same values)
    if (const int rc = check_attn_timeout(m, "decode_greedy")) {
        if (rc != PS_HIP_ATTN_TIMEOUT || retried) return rc;
        c->err.clear();
        return decode_greedy_impl(m, token, steps, out_ids, true);
    }
    m->position += (size_t)steps;
    return 0;
}

const float *ps_hip_model_logits(const ps_hip_model *m) { return m->logits; }
int ps_hip_model_argmax(ps_hip_model *m, int n, int32_t *ids_host) {
    ps_hip_ctx *c = m->ctx;
    if (n <= 0 || n > m->max_batch) PS_FAIL(c, "model_argmax: batch size out of range");
    if (int rc = settle_pending(m)) return rc;
    if (m->pin_argmax_n >= n) { // the last forward was a lowered one and left its ids in pinned memory
        if (m->pin_argmax_pending) { PS_CHECK(c, hipStreamSynchronize(c->stream)); m->pin_busy = false; m->pin_argmax_pending = false; }
        memcpy(ids_host, m->pin_argmax, (size_t)n * 4);
        return 0;
    }
    PS_CHECK(c, hipMemcpyAsync(ids_host, m->argmax_dev, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    m->pin_busy = false; m->pin_argmax_pending = false;
    return 0;
}
const float *ps_hip_model_scratch(const ps_hip_model *m, int which) {
    switch (which) {
    case 0: return m->x;      // [max_batch][dim]     residual stream after the last layer
    case 1: return m->q;      // [max_batch][dim]     rotated q of the last layer
    case 2: return m->att;    // [max_batch][dim]     attention output of the last layer
    case 3: return m->hb;     // [max_batch][hidden]  silu(gate)*up of the last layer
    case 4: return m->scores; // [max_batch][n_heads][n_ctx] raw scores of the last layer
    }
    return nullptr;
}
const float *ps_hip_model_k_cache(const ps_hip_model *m, int L) { return m->k_cache[L]; }
const float *ps_hip_model_v_cache(const ps_hip_model *m, int L) { return m->v_cache[L]; }
uint64_t ps_hip_model_weight_bytes_per_token(const ps_hip_model *m) { return m->weight_bytes; }
// Roofline helper for bench.py: replay mat-vec launches of one single-token forward exactly as the decode step issues
// them (same kernels, same fused prologues; every layer's own weights -> the stream is HBM-cold like a real step),
// `reps` times, bracketed by HIP events on the backend stream.  which = 0: every quantized mat-vec of a token;
// which = 1: only the gate/up launch of each layer (the dominant kernel).  Then the same number of empty launches to
// expose the launch boundary.
int ps_hip_model_bench_gemv(ps_hip_model *m, int reps, int which, double *seq_ms, double *null_ms, int *n_launches) {
    return ps_hip_model_bench_matmul(m, reps, which, 1, seq_ms, null_ms, n_launches);
}
// the same replay with `bs` activation columns (bs = 128: the mat-muls of a prefill chunk, quantizer launches included)
int ps_hip_model_bench_matmul(ps_hip_model *m, int reps, int which, int bs, double *seq_ms, double *null_ms, int *n_launches) {
    ps_hip_ctx *c = m->ctx;
    if (bs < 1 || bs > m->max_batch) PS_FAIL(c, "bench_matmul: batch size out of range");
    const ps_llm_config &f = m->cfg;
    const int64_t dim = f.dim, kvd = f.kv_dim, hid = f.hidden_dim;
    hipEvent_t e0, e1;
    PS_CHECK(c, hipEventCreate(&e0));
    PS_CHECK(c, hipEventCreate(&e1));
    int launches = 0;
    auto pass = [&](bool count) -> int {
        ps_act a1 = ps_act_carve(m->act_mem, dim, m->max_batch), a2 = ps_act_carve(m->act_mem, hid, m->max_batch);
        const bool qkv = which == 0 || which == 2, op = which == 0 || which == 3, gu = which == 0 || which == 1, dn = which == 0 || which == 4;
        for (uint32_t L = 0; L < f.n_layers; L++) {
            if (qkv) {
                psk_gemv_args g{};
                g.n_w = 3; g.w[0] = m->wq[L]; g.w[1] = m->wk[L]; g.w[2] = m->wv[L];
                g.out[0] = m->q; g.out[1] = m->k; g.out[2] = m->v; g.ldo[0] = dim; g.ldo[1] = kvd; g.ldo[2] = kvd;
                if (m->qwen2) { g.bias[0] = m->bq[L]; g.bias[1] = m->bk[L]; g.bias[2] = m->bv[L]; }
                g.pro = 1; g.pro_x = m->x; g.pro_norm_w = m->attn_norm[L]; g.pro_eps = f.norm_eps;
                if (mm(m, g, a1, dim, bs)) return 2;
                if (count) launches += 1;
            }
            if (op) {
                psk_gemv_args go{};
                go.n_w = 1; go.w[0] = m->wo[L]; go.out[0] = m->hb; go.ldo[0] = dim; // (not into x: the replay must not drift)
                go.pro = 2; go.pro_x = m->att;
                if (mm(m, go, a1, dim, bs)) return 2;
                if (count) launches += 1;
            }
            if (gu) {
                psk_gemv_args gf{};
                gf.n_w = 2; gf.w[0] = m->wg[L]; gf.w[1] = m->wu[L]; gf.out[0] = m->g1; gf.out[1] = m->g1; gf.ldo[0] = hid; gf.ldo[1] = hid; gf.silu_pair = 1;
                gf.pro = 1; gf.pro_x = m->x; gf.pro_norm_w = m->ffn_norm[L]; gf.pro_eps = f.norm_eps;
                if (mm(m, gf, a1, dim, bs)) return 2;
                if (count) launches += 1;
            }
            if (dn) {
                psk_gemv_args gd{};
                gd.n_w = 1; gd.w[0] = m->wd[L]; gd.out[0] = m->att; gd.ldo[0] = dim;
                gd.pro = 2; gd.pro_x = m->g1;
                if (mm(m, gd, a2, hid, bs)) return 2;
                if (count) launches += 1;
            }
        }
        if (which == 0 || which == 5) {
            psk_gemv_args gl{};
            gl.n_w = 1; gl.w[0] = m->output ? m->output : m->token_embd; gl.out[0] = m->logits; gl.ldo[0] = f.vocab_size;
            gl.pro = 1; gl.pro_x = m->x; gl.pro_norm_w = m->output_norm; gl.pro_eps = f.norm_eps;
            if (mm(m, gl, a1, dim, bs)) return 2;
            if (count) launches += 1;
        }
        return 0;
    };
    if (pass(true)) return 2; // warm-up + launch count
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    PS_CHECK(c, hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; r++) if (pass(false)) return 2;
    PS_CHECK(c, hipEventRecord(e1, c->stream));
    PS_CHECK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    PS_CHECK(c, hipEventElapsedTime(&ms, e0, e1));
    *seq_ms = (double)ms / reps;
    PS_CHECK(c, hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps * launches; r++) hipLaunchKernelGGL(null_kernel, dim3(256), dim3(256), 0, c->stream, (int *)nullptr);
    PS_CHECK(c, hipEventRecord(e1, c->stream));
    PS_CHECK(c, hipEventSynchronize(e1));
    PS_CHECK(c, hipEventElapsedTime(&ms, e0, e1));
    *null_ms = (double)ms / reps;
    *n_launches = launches;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

// PS_HIP_MODE_OR: mode bits OR-ed into every model's mode (A/B runs of unmodified drivers, e.g. bench.py with 16 = two-launch attention)
static int mode_env_or() {
    static const int v = [] { const char *e = getenv("PS_HIP_MODE_OR"); return e ? atoi(e) : 0; }();
    return v;
}
int ps_hip_model_set_mode(ps_hip_model *m, int mode) {
    mode |= mode_env_or();
    if (mode & 64) { m->attn1_disabled = false; mode &= ~64; } // explicit re-arm of the one-launch attention after a time-out
    if (m->attn1_disabled) mode |= 16;                         // sticky: a later set_mode without bit 4 does not bring the form that timed out back
    if ((mode & 8) && m->k16.empty()) { // fp16 mirrors of the caches: filled from now on, so the cache must be empty
        if (m->position != 0) { m->ctx->err = "set_mode: the fp16-KV decode mode must be switched on while the cache is empty"; return 2; }
        const size_t n = (size_t)m->cfg.seq_len * m->cfg.kv_dim * 2;
        m->k16.assign(m->cfg.n_layers, nullptr); m->v16.assign(m->cfg.n_layers, nullptr);
        for (uint32_t i = 0; i < m->cfg.n_layers; i++) {
            if (dmalloc(m, (void **)&m->k16[i], n) || dmalloc(m, (void **)&m->v16[i], n)) { m->k16.clear(); m->v16.clear(); return 2; }
            (void)hipMemsetAsync(m->k16[i], 0, n, m->ctx->stream);
            (void)hipMemsetAsync(m->v16[i], 0, n, m->ctx->stream);
        }
        if (dmalloc(m, (void **)&m->attn_part, (size_t)m->cfg.n_heads * 32 * (m->cfg.head_size + 2) * 4)) return 2;
    }
    if ((mode & 8) && !(m->mode & 8) && m->position != 0) { m->ctx->err = "set_mode: the fp16-KV decode mode must be switched on while the cache is empty"; return 2; }
    if ((m->mode ^ mode) & (62 | 128)) drop_graphs(m); // the captured steps bake the launch plan in
    m->mode = mode;
    return 0;
}

} // extern "C"
