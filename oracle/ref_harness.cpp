// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI harness (my code) around the REAL reference compiled in place from /root/reference
// (see oracle/Makefile).  It exposes
//   (a) the op-level entry points the reference's ggml backend calls —
//       powerserve_compute_forward_{mul_mat,rms_norm,rope,softmax_ext,add,dup}
//       (libs/ggml/include/ggml.h:764-817) — fanned out on the reference's own ThreadPool exactly
//       like GGMLBackend::matmul does (src/backend/ggml/ggml_wrapper.cpp:20-40), and
//   (b) the end-to-end LlamaModel/Qwen2Model::forward (src/model/llama/llama_model.cpp:52-117)
//       on a GGUF file, with greedy arg-max sampling (top_k=1 == max_element,
//       src/sampler/prob_array.cpp:65-67), and
//   (c) the reference's sampler classes chained in the order of SamplerChain::build_from_config
//       (src/sampler/sampler_chain.cpp:19-51; built with append<> because build_from_config wants a Tokenizer).
// Nothing here is shipped in the product; tests/, smoke() and bench.py's cpu_baseline leg load the
// resulting oracle/_ref/libps_ref.so through ctypes.

#include "backend/cpu_buffer.hpp"
#include "backend/ggml/ggml.hpp"
#include "backend/platform.hpp"
#include "core/config.hpp"
#include "core/thread_pool.hpp"
#include "ggml-quants.h"
#include "ggml.h"
#include "model/llama/llama_model.hpp"
#include "model/module/norm_attention.hpp"
#include "model/qwen2/qwen2_model.hpp"
#include "sampler/sampler_chain.hpp"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace powerserve;

extern "C" {

struct ref_tensor {
    int32_t type;  // enum ggml_type value
    int32_t _pad;
    int64_t ne[4]; // elements per dim, dim0 fastest
    uint64_t nb[4]; // strides in BYTES (ggml convention)
    void *data;
};

struct ref_rope_params {
    int32_t n_dims, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
    int32_t mode;
};

struct ref_llm_config {
    uint32_t dim, hidden_dim, n_layers, n_heads, n_kv_heads, seq_len, vocab_size, kv_dim, head_size;
    float norm_eps;
    ref_rope_params rope;
};

} // extern "C"

namespace {

struct RefCtx {
    int n_threads;
    std::unique_ptr<ThreadPool> pool;
    std::vector<char> wdata;
    std::atomic<int> chunk{0};
};

void ensure_ggml_init() {
    static bool done = false;
    if (!done) {
        // fills the fp16->fp32 table every x86 GGML_FP16_TO_FP32 goes through (ggml.c:3690-3715)
        struct ggml_init_params p = {/*mem_size*/ 1 << 20, /*mem_buffer*/ nullptr, /*no_alloc*/ true};
        struct ggml_context *c    = ggml_init(p);
        (void)c; // intentionally kept alive
        done = true;
    }
}

void to_ggml(const ref_tensor *t, ggml_tensor *g) {
    memset(g, 0, sizeof(*g));
    g->type = (enum ggml_type)t->type;
    g->data = t->data;
    for (int i = 0; i < 4; i++) {
        g->ne[i] = t->ne[i];
        g->nb[i] = t->nb[i];
    }
}

template <typename F>
void run_on_pool(RefCtx *c, F &&fn) {
    c->pool->run([&](size_t tid) {
        op_compute_params p{};
        p.ith           = (int)tid;
        p.nth           = (int)c->pool->size();
        p.wsize         = c->wdata.size();
        p.wdata         = c->wdata.data();
        p.thread_pool   = (void *)c->pool.get();
        p.barrier_fn    = [](void *o) { ((ThreadPool *)o)->barrier(); };
        p.current_chunk = (atomic_int *)&c->chunk;
        fn(&p);
    });
}

void need_wdata(RefCtx *c, size_t bytes) {
    bytes += (size_t)get_cache_line_size() * c->n_threads + 4096;
    if (c->wdata.size() < bytes) c->wdata.resize(bytes);
}

} // namespace

extern "C" {

// ---------------------------------------------------------------- type helpers
size_t ref_row_size(int type, int64_t k) {
    return ggml_row_size((enum ggml_type)type, k);
}
int64_t ref_blck_size(int type) {
    return ggml_blck_size((enum ggml_type)type);
}
size_t ref_type_size(int type) {
    return ggml_type_size((enum ggml_type)type);
}
int ref_vec_dot_type(int type) {
    ggml_tensor t{};
    t.type = (enum ggml_type)type;
    return (int)powerserve_get_vec_dot_type(&t);
}
// weights: float rows -> quant blocks (ggml_quantize_chunk, ggml.c:22961)
void ref_quantize_chunk(int type, const float *src, void *dst, int64_t nrows, int64_t k) {
    ensure_ggml_init();
    ggml_quantize_chunk((enum ggml_type)type, src, dst, 0, nrows, k, nullptr);
}
void ref_dequantize_row(int type, const void *src, float *dst, int64_t k) {
    ensure_ggml_init();
    ggml_internal_get_type_traits((enum ggml_type)type).to_float(src, dst, k);
}
// activation quantizer the mat-mul uses (type_traits[vec_dot_type].from_float, ggml.c:13502-13530)
void ref_from_float(int type, const float *src, void *dst, int64_t k) {
    ensure_ggml_init();
    ggml_internal_get_type_traits((enum ggml_type)type).from_float(src, dst, k);
}

// ---------------------------------------------------------------- op-level context
void *ref_ctx_create(int n_threads) {
    ensure_ggml_init();
    auto c       = new RefCtx();
    c->n_threads = n_threads;
    std::vector<ThreadConfig> cfg(n_threads);
    c->pool = std::make_unique<ThreadPool>(cfg);
    return c;
}
void ref_ctx_destroy(void *h) {
    delete (RefCtx *)h;
}

// dst = src0(weight or K/V view) x src1(activation).  If act_out != NULL the quantized activation
// rows the reference wrote into wdata are copied out (fixture F1).
int ref_mul_mat(void *h, const ref_tensor *dst, const ref_tensor *src0, const ref_tensor *src1, void *act_out,
                size_t act_cap) {
    auto c = (RefCtx *)h;
    ggml_tensor d, a, b;
    to_ggml(dst, &d);
    to_ggml(src0, &a);
    to_ggml(src1, &b);
    const enum ggml_type vdt = powerserve_get_vec_dot_type(&a);
    size_t ws                = 0;
    if (b.type != vdt) {
        // same sizing as GGMLBackend::plan (src/backend/ggml/ggml.cpp:61-69)
        ws = ggml_row_size(vdt, b.ne[0] * b.ne[1] * b.ne[2] * b.ne[3]);
    }
    need_wdata(c, ws);
    run_on_pool(c, [&](op_compute_params *p) { powerserve_compute_forward_mul_mat(p, &d, &a, &b); });
    if (act_out && ws) {
        if (act_cap < ws) return -1;
        memcpy(act_out, c->wdata.data(), ws);
    }
    return 0;
}

int ref_rms_norm(void *h, const ref_tensor *dst, const ref_tensor *src0, const ref_tensor *w, float eps) {
    auto c = (RefCtx *)h;
    ggml_tensor d, a, b;
    to_ggml(dst, &d);
    to_ggml(src0, &a);
    to_ggml(w, &b);
    run_on_pool(c, [&](op_compute_params *p) { powerserve_compute_forward_rms_norm(p, &d, &a, &b, eps); });
    return 0;
}

int ref_rope(void *h, const ref_tensor *dst, const ref_tensor *src0, const int32_t *pos, int npos,
             const ref_rope_params *rp) {
    auto c = (RefCtx *)h;
    ggml_tensor d, a, pt;
    to_ggml(dst, &d);
    to_ggml(src0, &a);
    memset(&pt, 0, sizeof(pt));
    pt.data  = (void *)pos;
    pt.type  = GGML_TYPE_I32;
    pt.ne[0] = npos;
    pt.ne[1] = pt.ne[2] = pt.ne[3] = 1;
    pt.nb[0]                       = 4;
    pt.nb[1] = pt.nb[2] = pt.nb[3] = 4 * (size_t)npos;
    rope_compute_params r{rp->n_dims,      rp->n_ctx_orig, rp->freq_base, rp->freq_scale, rp->ext_factor,
                          rp->attn_factor, rp->beta_fast,  rp->beta_slow, rp->mode};
    need_wdata(c, sizeof(float) * (size_t)(d.ne[0] + 64) * c->n_threads);
    run_on_pool(c, [&](op_compute_params *p) { powerserve_compute_forward_rope(p, &d, &a, &pt, nullptr, &r); });
    return 0;
}

int ref_softmax_ext(void *h, const ref_tensor *dst, const ref_tensor *src0, const ref_tensor *mask, float scale,
                    float max_bias) {
    auto c = (RefCtx *)h;
    ggml_tensor d, a, m;
    to_ggml(dst, &d);
    to_ggml(src0, &a);
    to_ggml(mask, &m);
    need_wdata(c, sizeof(float) * (size_t)(d.ne[0] + 64) * c->n_threads);
    run_on_pool(c, [&](op_compute_params *p) {
        // n_tasks = min(n_threads, nrows)  (GGMLBackend::get_n_tasks, ggml_wrapper.cpp:264-267) — rows are
        // partitioned by (ith, nth) inside, extra threads simply get an empty range.
        powerserve_compute_forward_softmax_ext(p, &d, &a, &m, scale, max_bias);
    });
    return 0;
}

int ref_add(void *h, const ref_tensor *dst, const ref_tensor *a_, const ref_tensor *b_) {
    auto c = (RefCtx *)h;
    ggml_tensor d, a, b;
    to_ggml(dst, &d);
    to_ggml(a_, &a);
    to_ggml(b_, &b);
    run_on_pool(c, [&](op_compute_params *p) { powerserve_compute_forward_add(p, &d, &a, &b); });
    return 0;
}

int ref_dup(void *h, const ref_tensor *dst, const ref_tensor *src) {
    auto c = (RefCtx *)h;
    ggml_tensor d, a;
    to_ggml(dst, &d);
    to_ggml(src, &a);
    run_on_pool(c, [&](op_compute_params *p) { powerserve_compute_forward_dup(p, &d, &a); });
    return 0;
}

// GGMLBackend::silu_hadamard (src/backend/ggml/ggml.cpp:115-129) and ::get_embedding
// (ggml_wrapper.cpp:181-211) are C++ members, called here on a real backend object.
struct RefBackendBox {
    ModelConfig::LLMConfig cfg;
    HyperParams hp;
    std::unique_ptr<ggml::GGMLBackend> be;
};
static RefBackendBox *tiny_backend() {
    static RefBackendBox *box = nullptr;
    if (!box) {
        box                 = new RefBackendBox();
        box->cfg.dim        = 32;
        box->cfg.n_layers   = 1;
        box->cfg.n_heads    = 1;
        box->cfg.n_kv_heads = 1;
        box->cfg.seq_len    = 8;
        box->cfg.kv_dim     = 32;
        box->cfg.head_size  = 32;
        box->hp.n_threads   = 1;
        box->be             = std::make_unique<ggml::GGMLBackend>(box->cfg, box->hp);
    }
    return box;
}
static Tensor mk_f32(float *p, size_t n0, size_t n1) {
    Tensor t(DataType::FP32, {n0, n1, 1, 1});
    Stride s{4, 4 * n0, 4 * n0 * n1, 4 * n0 * n1};
    t.m_data = std::make_shared<CPUBuffer>(s, (void *)p);
    return t;
}
int ref_silu_hadamard(float *out, const float *gate, const float *up, int64_t n0, int64_t n1) {
    ensure_ggml_init();
    auto o = mk_f32(out, n0, n1), g = mk_f32((float *)gate, n0, n1), u = mk_f32((float *)up, n0, n1);
    tiny_backend()->be->silu_hadamard(&o, &g, &u);
    return 0;
}
// weight: [dim, vocab] rows of ggml type `type` (F32 / Q4_0 / Q8_0 only — the reference aborts otherwise)
int ref_get_embedding(int type, const void *table, int64_t dim, int64_t vocab, const int32_t *tokens, int n,
                      float *out) {
    ensure_ggml_init();
    DataType dt;
    switch (type) {
    case GGML_TYPE_F32: dt = DataType::FP32; break;
    case GGML_TYPE_Q4_0: dt = DataType::GGML_Q4_0; break;
    case GGML_TYPE_Q8_0: dt = DataType::GGML_Q8_0; break;
    default: return -1;
    }
    Tensor w(dt, {(size_t)dim, (size_t)vocab, 1, 1});
    size_t rs = ggml_row_size((enum ggml_type)type, dim);
    Stride s{ggml_type_size((enum ggml_type)type), rs, rs * vocab, rs * vocab};
    w.m_data = std::make_shared<CPUBuffer>(s, (void *)table);
    auto o   = mk_f32(out, dim, n);
    std::vector<int> toks(tokens, tokens + n);
    tiny_backend()->be->get_embedding(&o, &w, toks);
    return 0;
}

// ---------------------------------------------------------------- end-to-end model
struct RefModel {
    std::shared_ptr<ModelConfig> cfg;
    std::shared_ptr<Model> model;
    std::shared_ptr<Platform> platform;
    ggml::GGMLBackend *be = nullptr;
    bool pool_up          = false;
};

void *ref_model_create(const char *gguf_path, const char *arch, const ref_llm_config *lc, int n_threads) {
    ensure_ggml_init();
    auto m           = new RefModel();
    m->cfg           = std::make_shared<ModelConfig>();
    m->cfg->version  = 1;
    m->cfg->arch     = arch;
    m->cfg->model_id = std::string("ref_") + arch;
    auto &l          = m->cfg->llm;
    l.dim            = lc->dim;
    l.hidden_dim     = lc->hidden_dim;
    l.n_layers       = lc->n_layers;
    l.n_heads        = lc->n_heads;
    l.n_kv_heads     = lc->n_kv_heads;
    l.seq_len        = lc->seq_len;
    l.vocab_size     = lc->vocab_size;
    l.kv_dim         = lc->kv_dim;
    l.head_size      = lc->head_size;
    l.norm_eps       = lc->norm_eps;
    auto &r          = l.rope_config;
    r.n_dims         = lc->rope.n_dims;
    r.n_ctx_orig     = lc->rope.n_ctx_orig;
    r.freq_base      = lc->rope.freq_base;
    r.freq_scale     = lc->rope.freq_scale;
    r.ext_factor     = lc->rope.ext_factor;
    r.attn_factor    = lc->rope.attn_factor;
    r.beta_fast      = lc->rope.beta_fast;
    r.beta_slow      = lc->rope.beta_slow;
    r.rope_type      = lc->rope.mode;

    if (std::string(arch) == "llama") {
        m->model = std::make_shared<LlamaModel>(gguf_path, m->cfg);
    } else if (std::string(arch) == "qwen2") {
        m->model = std::make_shared<Qwen2Model>(gguf_path, m->cfg);
    } else {
        delete m;
        return nullptr;
    }
    m->platform           = std::make_shared<Platform>();
    m->model->m_platform  = m->platform;
    HyperParams hp;
    hp.n_threads = n_threads;
    m->platform->init_ggml_backend(m->cfg, hp);
    m->model->m_attn = std::make_shared<NormAttention>(m->cfg->llm, m->model->m_weights);
    m->be            = m->platform->ggml_backends[m->cfg->model_id].get();
    m->be->setup_threadpool();
    m->pool_up = true;
    // Work data for the longest soft-max row up front.  GGMLBackend::setup_work_data (src/backend/ggml/ggml.cpp:99-109) compares the request WITHOUT its
    // cache-line pad against the buffer WITH it, so a soft-max whose 4 * n_kv * n_threads bytes fall within the last 64 * n_threads bytes of the current
    // buffer is not given its per-thread pad and ggml_compute_forward_soft_max_f32 (ggml.c:14905) writes past the vector: at n_kv = 77 with two threads
    // on the 256-wide test models (AddressSanitizer: heap-buffer-overflow, 0 bytes to the right of a 672-byte region), i.e. any single-token sequence of a
    // few dozen steps corrupts the heap (found by tools/cpu_fuzz_oracle.py; the reference's own models only get there at n_kv of a thousand and more,
    // where the allocator's slack usually absorbs it).  Scratch only: no result depends on the buffer's size.
    // (the same comparison bites ROPE's per-thread cache, ggml.c:15349, when the buffer is SMALLER than a mat-mul would have made it: ask for well above both)
    m->be->setup_work_data(4 * ((size_t)l.seq_len + (size_t)l.head_size + 32) * (size_t)n_threads + 65536);
    return m;
}

void ref_model_destroy(void *h) {
    auto m = (RefModel *)h;
    if (m->pool_up) m->be->reset_threadpool();
    delete m;
}

size_t ref_model_kv_position(void *h) {
    auto m = (RefModel *)h;
    return m->platform->get_kv_position(m->cfg->model_id);
}

void ref_model_reset(void *h) {
    auto m = (RefModel *)h;
    m->platform->reset_kv_position(m->cfg->model_id);
}

// one Model::forward; logits_out (may be NULL) receives [n][vocab] floats when lm_head != 0
int ref_model_forward(void *h, const int32_t *tokens, int n, const int32_t *pos, int lm_head, float *logits_out) {
    auto m = (RefModel *)h;
    std::vector<int> t(tokens, tokens + n), p(pos, pos + n);
    CausalAttentionMask mask(n);
    auto ret = m->model->forward(t, p, mask, lm_head != 0);
    if (lm_head && logits_out) {
        size_t v = m->cfg->llm.vocab_size;
        for (int i = 0; i < n; i++) memcpy(logits_out + (size_t)i * v, ret.logits_vector[i].data(), v * sizeof(float));
    }
    return 0;
}

// ModelTokenIterator semantics (src/model/model.hpp:117-184): prefill all but the last prompt token in
// chunks of batch_size with lm_head=false, then `steps` single-token greedy steps starting from the last
// prompt token.  out_tokens[steps]; logits_out (may be NULL) gets [steps][vocab].
int ref_model_generate(void *h, const int32_t *prompt, int n_prompt, int batch_size, int steps, int32_t *out_tokens,
                       float *logits_out, double *t_prefill_s, double *t_decode_s) {
    auto m = (RefModel *)h;
    using clk = std::chrono::steady_clock;
    ref_model_reset(h);
    size_t position = ref_model_kv_position(h);
    auto t0         = clk::now();
    int n_prefilled = 0;
    while (n_prefilled < n_prompt - 1) {
        int bs = std::min(batch_size, n_prompt - n_prefilled - 1);
        std::vector<int> t(prompt + n_prefilled, prompt + n_prefilled + bs), p(bs);
        for (int i = 0; i < bs; i++) p[i] = (int)position + i;
        CausalAttentionMask mask(bs);
        m->model->forward(t, p, mask, false);
        position = ref_model_kv_position(h);
        n_prefilled += bs;
    }
    auto t1   = clk::now();
    int cur   = prompt[n_prompt - 1];
    size_t v  = m->cfg->llm.vocab_size;
    for (int s = 0; s < steps; s++) {
        std::vector<int> t(1, cur), p(1, (int)ref_model_kv_position(h));
        CausalAttentionMask mask(1);
        auto ret          = m->model->forward(t, p, mask, true);
        const float *lg   = ret.logits_vector[0].data();
        size_t best       = 0;
        for (size_t i = 1; i < v; i++)
            if (lg[i] > lg[best]) best = i;
        if (logits_out) memcpy(logits_out + (size_t)s * v, lg, v * sizeof(float));
        out_tokens[s] = (int32_t)best;
        cur           = (int)best;
    }
    auto t2 = clk::now();
    if (t_prefill_s) *t_prefill_s = std::chrono::duration<double>(t1 - t0).count();
    if (t_decode_s) *t_decode_s = std::chrono::duration<double>(t2 - t1).count();
    return 0;
}

} // extern "C"

// ---------------------------------------------------------------- (c) the reference's sampler chain
extern "C" {
struct ref_sampler_cfg { // same layout as psh_sampler_cfg (powerserve_amd/csrc/host/model.cpp)
    uint64_t seed;
    float temperature, top_p;
    uint64_t top_k;
    int32_t penalty_last_n;
    float penalty_repeat, penalty_freq, penalty_present;
    int32_t penalize_nl, ignore_eos, n_vocabs, special_eos_id, linefeed_id;
};
void *ref_sampler_create(const ref_sampler_cfg *c) {
    auto *ch = new SamplerChain();
    ch->append<RepeatPenaltySampler>(c->n_vocabs, c->special_eos_id, c->linefeed_id, c->penalty_last_n, c->penalty_repeat, c->penalty_freq,
                                     c->penalty_present, c->penalize_nl != 0, c->ignore_eos != 0);
    ch->append<TopKSampler>((size_t)c->top_k);
    ch->append<TemperatureSampler>(c->temperature);
    ch->append<SoftmaxSampler>();
    ch->append<TopPSampler>(c->top_p);
    ch->append<NormalizeSampler>();
    ch->append<StochasticSampler>(c->seed);
    return ch;
}
void ref_sampler_free(void *s) { delete (SamplerChain *)s; }
int32_t ref_sampler_sample(void *s, const float *logits, int n) { // ModelTokenIterator::decode: apply, probs[0], accept
    auto *ch = (SamplerChain *)s;
    ProbArray probs(std::span<const float>(logits, (size_t)n));
    ch->apply(probs);
    const Token next = probs[0].token;
    ch->accept(next);
    return next;
}
}
