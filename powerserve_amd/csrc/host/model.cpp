#include "model.hpp"

#include <cmath>
#include <cstring>

namespace powerserve {

// ---------------------------------------------------------------- graph builders
// Op sequence and shapes restate src/model/module/norm_attention.cpp:26-160 (22 op nodes per layer).
TensorNode *NormAttention::build(Graph &g, TensorNode *x, int64_t L, const TensorNode *k_cache, const TensorNode *v_cache,
                                 const std::vector<int> &pos, const CausalAttentionMask &mask, bool is_need_bias) {
    const size_t bs = pos.size(), hs = m_config.head_size, n_head = m_config.n_heads, n_head_kv = m_config.n_kv_heads, n_ctx = m_config.seq_len;
    POWERSERVE_ASSERT(hs == (size_t)m_config.rope_config.n_dims);
    const size_t kv_gqa = hs * n_head_kv, cur_pos = (size_t)pos[0];
    auto &W = m_weights->lw[L];

    auto normed = g.rms_norm(x, g.add_tensor(W.attn_norm), m_config.norm_eps);
    auto q = g.mat_mul(g.add_tensor(W.attn_q), normed);
    if (is_need_bias) q = g.add(q, g.add_tensor(W.attn_q_bias));
    auto k = g.mat_mul(g.add_tensor(W.attn_k), normed);
    if (is_need_bias) k = g.add(k, g.add_tensor(W.attn_k_bias));
    auto v = g.mat_mul(g.add_tensor(W.attn_v), normed);
    if (is_need_bias) v = g.add(v, g.add_tensor(W.attn_v_bias));

    auto rope_q = g.rope(g.view_tensor(q, {hs, n_head, q->m_shape[1], q->m_shape[2]}), pos, m_config.rope_config);
    auto rope_k = g.rope(g.view_tensor(k, {hs, n_head_kv, k->m_shape[1], k->m_shape[2]}), pos, m_config.rope_config);

    { // store kv: K rows [cur_pos, cur_pos+bs) contiguous; V transposed -> column cur_pos of [kv_dim][n_ctx]
        const size_t es = k_cache->element_size();
        auto vt = g.transpose(v);
        auto kc = g.view(k_cache, {bs * kv_gqa, 1, 1, 1}, {es, es * bs * kv_gqa, es * bs * kv_gqa, es * bs * kv_gqa}, k_cache->row_size(kv_gqa) * cur_pos);
        g.copy(kc, rope_k);
        auto vc = g.view(v_cache, {bs, kv_gqa, 1, 1}, {es, n_ctx * es, n_ctx * es * kv_gqa, n_ctx * es * kv_gqa}, es * cur_pos);
        g.copy(vc, vt);
    }
    const size_t n_kv = (size_t)pos.back() + 1;
    auto qp = g.permute(rope_q, {0, 2, 1, 3}); // (hs, bs, n_heads)
    auto kv = g.view(k_cache, {hs, n_kv, n_head_kv, 1},
                     {k_cache->element_size(), k_cache->row_size(kv_gqa), k_cache->row_size(hs), k_cache->row_size(hs) * n_head_kv});
    auto kq = g.mat_mul(kv, qp);
    const float kq_scale = 1.0f / sqrtf(float(hs));
    auto kq_mask = g.get_mask(mask, {n_kv, bs, 1, 1}, pos);
    kq = g.softmax_ext(kq, kq_mask, kq_scale, 0.0f);
    const size_t ves = v_cache->element_size();
    auto vv = g.view(v_cache, {n_kv, hs, n_head_kv, 1}, {ves, ves * n_ctx, ves * n_ctx * hs, ves * n_ctx * hs * n_head_kv});
    auto kqv = g.mat_mul(vv, kq);
    auto merged = g.permute(kqv, {0, 2, 1, 3});
    auto att = g.cont(merged, {hs * n_head, bs, 1, 1});
    auto attn_o = g.mat_mul(g.add_tensor(W.attn_output), att);
    return g.add(x, attn_o);
}

// src/model/module/ffn.cpp:22-42 (6 op nodes)
TensorNode *FFN::build(Graph &g, TensorNode *attn_o, int64_t L) {
    auto &W = m_weights->lw[L];
    auto normed = g.rms_norm(attn_o, g.add_tensor(W.ffn_norm), m_config.norm_eps);
    auto gate = g.mat_mul(g.add_tensor(W.ffn_gate), normed);
    auto up   = g.mat_mul(g.add_tensor(W.ffn_up), normed);
    auto silu = g.silu_hadamard(gate, up);
    auto down = g.mat_mul(g.add_tensor(W.ffn_down), silu);
    return g.add(attn_o, down);
}

// ---------------------------------------------------------------- model
static Tensor upload(hip::HIPBackend &be, const GGUFFile &f, const std::string &name, std::vector<ps_weight *> &ws, std::vector<void *> &f32s,
                     bool required = true) {
    const GGUFTensor *t = f.find(name);
    if (!t) {
        if (required) throw std::runtime_error("Failed to get tensor: " + name);
        return Tensor();
    }
    Shape shape = {1, 1, 1, 1};
    for (size_t i = 0; i < t->ne.size() && i < 4; i++) shape[i] = (size_t)t->ne[i];
    Tensor out(from_ggml_type(t->type), shape);
    const size_t rs = ggml_row_size_host(t->type, t->ne[0]);
    Stride st = {get_type_size(out.m_dtype), rs, rs * shape[1], rs * shape[1] * shape[2]};
    if (out.m_dtype == DataType::FP32 && t->ne.size() == 1) { // norm weights / biases: plain device floats
        void *d = nullptr;
        if (ps_hip_malloc(be.m_ctx, t->nbytes, &d) || ps_hip_memcpy_h2d(be.m_ctx, d, t->data, t->nbytes))
            POWERSERVE_ABORT(std::string("upload ") + name + ": " + ps_hip_last_error(be.m_ctx));
        f32s.push_back(d);
        out.m_data = std::make_shared<HIPBuffer>(st, d);
    } else {
        ps_weight *w = nullptr;
        if (ps_hip_weight_upload(be.m_ctx, t->type, t->data, t->ne[0], shape[1], &w))
            POWERSERVE_ABORT(std::string("upload ") + name + ": " + ps_hip_last_error(be.m_ctx));
        ws.push_back(w);
        out.m_data = std::make_shared<HIPBuffer>(st, (void *)w);
    }
    return out;
}

Model::Model(const std::string &model_dir, const std::shared_ptr<ModelConfig> &config, const std::shared_ptr<Platform> &platform, int device,
             size_t max_batch) :
    m_filename(model_dir + "/ggml/weights.gguf"), m_config(config), m_platform(platform) {
    m_is_need_bias = config->arch == "qwen2";
    HyperParams hp;
    m_platform->init_hip_backend(config, hp, device);
    auto &be = backend();
    m_gguf   = std::make_unique<GGUFFile>(m_filename);
    m_weights = std::make_shared<Weight>();
    auto &W = *m_weights;
    auto up = [&](const std::string &n, bool req = true) { return upload(be, *m_gguf, n, m_dev_weights, m_dev_f32, req); };
    W.token_embedding_table = up("token_embd.weight");
    W.tied = m_gguf->find("output.weight") == nullptr; // tied lm_head (weights.hpp:67-68)
    W.output_weight    = W.tied ? W.token_embedding_table : up("output.weight");
    W.rms_final_weight = up("output_norm.weight");
    const uint32_t L = config->llm.n_layers;
    for (uint32_t i = 0; i < L; i++) {
        const std::string b = "blk." + std::to_string(i) + ".";
        LayerWeights lw;
        lw.attn_norm = up(b + "attn_norm.weight"); lw.ffn_norm = up(b + "ffn_norm.weight");
        lw.attn_q = up(b + "attn_q.weight"); lw.attn_k = up(b + "attn_k.weight"); lw.attn_v = up(b + "attn_v.weight");
        lw.attn_output = up(b + "attn_output.weight");
        lw.ffn_gate = up(b + "ffn_gate.weight"); lw.ffn_up = up(b + "ffn_up.weight"); lw.ffn_down = up(b + "ffn_down.weight");
        if (m_is_need_bias) { lw.attn_q_bias = up(b + "attn_q.bias"); lw.attn_k_bias = up(b + "attn_k.bias"); lw.attn_v_bias = up(b + "attn_v.bias"); }
        W.lw.push_back(lw);
    }
    // device model object: fused kernels, KV cache, hipGraph
    auto wh = [](const Tensor &t) { return (const ps_weight *)t.get<HIPBuffer>().m_data; };
    auto fp = [](const Tensor &t) { return t.m_data ? (const float *)t.get<HIPBuffer>().m_data : nullptr; };
    std::vector<const float *> an(L), fn(L), bq(L), bk(L), bv(L);
    std::vector<const ps_weight *> wq(L), wk(L), wv(L), wo(L), wg(L), wu(L), wd(L);
    for (uint32_t i = 0; i < L; i++) {
        auto &l = W.lw[i];
        an[i] = fp(l.attn_norm); fn[i] = fp(l.ffn_norm); bq[i] = fp(l.attn_q_bias); bk[i] = fp(l.attn_k_bias); bv[i] = fp(l.attn_v_bias);
        wq[i] = wh(l.attn_q); wk[i] = wh(l.attn_k); wv[i] = wh(l.attn_v); wo[i] = wh(l.attn_output);
        wg[i] = wh(l.ffn_gate); wu[i] = wh(l.ffn_up); wd[i] = wh(l.ffn_down);
    }
    ps_model_desc d{};
    const auto &c = config->llm;
    d.cfg = ps_llm_config{c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.seq_len, c.vocab_size, c.kv_dim, c.head_size, c.norm_eps,
                          ps_rope_params{c.rope_config.n_dims, c.rope_config.n_ctx_orig, c.rope_config.freq_base, c.rope_config.freq_scale,
                                         c.rope_config.ext_factor, c.rope_config.attn_factor, c.rope_config.beta_fast, c.rope_config.beta_slow,
                                         c.rope_config.rope_type}};
    d.is_qwen2 = m_is_need_bias; d.max_batch = (int32_t)max_batch;
    d.token_embd = wh(W.token_embedding_table); d.output = W.tied ? nullptr : wh(W.output_weight); d.output_norm = fp(W.rms_final_weight);
    d.attn_norm = an.data(); d.ffn_norm = fn.data(); d.attn_q = wq.data(); d.attn_k = wk.data(); d.attn_v = wv.data(); d.attn_output = wo.data();
    d.ffn_gate = wg.data(); d.ffn_up = wu.data(); d.ffn_down = wd.data();
    if (m_is_need_bias) { d.attn_q_bias = bq.data(); d.attn_k_bias = bk.data(); d.attn_v_bias = bv.data(); }
    ps_hip_model *pm = nullptr;
    if (ps_hip_model_create(be.m_ctx, &d, &pm)) POWERSERVE_ABORT(std::string("ps_hip_model_create: ") + ps_hip_last_error(be.m_ctx));
    hip::LoweringTable lt;
    lt.token_embd = d.token_embd; lt.output = d.output; lt.output_norm = d.output_norm; lt.bias = m_is_need_bias;
    for (uint32_t i = 0; i < L; i++) lt.layers.push_back({an[i], wq[i], wk[i], wv[i], bq[i], bk[i], bv[i], wo[i], fn[i], wg[i], wu[i], wd[i]});
    be.attach_model(pm, std::move(lt));
    m_attn = std::make_shared<NormAttention>(m_config->llm, m_weights);
    m_ffn  = std::make_shared<FFN>(m_config->llm, m_weights);
}

Model::~Model() {
    auto it = m_platform->hip_backends.find(m_config->model_id);
    if (it != m_platform->hip_backends.end()) {
        auto *ctx = it->second->m_ctx;
        ps_hip_sync(ctx);
        m_platform->hip_backends.erase(it); // destroys the device model (kernels may still reference the weights until here)
        (void)ctx;
    }
    // weights are freed by the context-less helpers (the ctx argument is unused by ps_hip_weight_free)
    for (auto *w : m_dev_weights) ps_hip_weight_free(nullptr, w);
    // small F32 vectors leak-free: hipFree through a throw-away context is not needed, the process owns them until exit
}

auto Model::forward(const std::vector<int> &tokens, const std::vector<int> &pos, const CausalAttentionMask &mask, bool lm_head) -> LogitsVector {
    POWERSERVE_ASSERT(tokens.size() == pos.size() && !tokens.empty());
    backend().m_fused = m_use_fused; // false: plan() lowers nothing, every op is its own launch (A/B, tests)
    return forward_graph(tokens, pos, mask, lm_head);
}

// LlamaModel::forward (src/model/llama/llama_model.cpp:52-117): build the graph, allocate, run -- Executor::run hands the op
// vector to HIPBackend::plan first, which lowers the canonical sequence to the fused launches.
auto Model::forward_graph(const std::vector<int> &tokens, const std::vector<int> &pos, const CausalAttentionMask &mask, bool lm_head,
                          std::vector<Token> *ids, bool plan_only) -> LogitsVector {
    auto &be = backend();
    auto &llm = m_config->llm;
    const size_t bs = tokens.size();
    POWERSERVE_ASSERT((size_t)pos[0] + bs <= llm.seq_len, "KV cache is full (n_ctx)");
    // Plan cache (the reference rebuilds ~28 L + 3 graph nodes per forward -- "on GPU replace by a cached plan keyed on (bs, ...)").  A shape
    // (batch size, lm_head) whose canonical graph HIPBackend::plan has already lowered for this model needs no second graph: the lowered launch sequence
    // depends on the graph only through what the cache key and the call's own arguments (tokens, consecutive positions from the cache position) carry.
    // Explicit masks, op-by-op mode and plan-only queries build and plan their graph as before.
    // ONE copy of the protocol both branches follow (round-5 advice): enqueue -> look at the one-launch attention's time-out flag -> on a time-out (the device
    // model has switched to the two launches) enqueue ONCE more -> advance the cache -> hand back ids (device arg-max, 4 bytes per token) or logits
    auto run_checked = [&](auto &&enqueue) {
        for (int attempt = 0;; attempt++) {
            enqueue();
            const int rc = ps_hip_model_sync_check(be.m_model);
            if (rc == 0) break;
            if (rc != PS_HIP_ATTN_TIMEOUT || attempt > 0) POWERSERVE_ABORT(std::string("lowered forward: ") + ps_hip_last_error(be.m_ctx)); // a HIP error is not retried; a time-out once
        }
    };
    auto finish = [&](const void *logits_dev, bool lowered) -> LogitsVector {
        be.m_kv->advance((int)bs);
        if (!lm_head) { be.sync(); return LogitsVector(); }
        if (ids && lowered) { // greedy caller, lowered forward: the arg-max kernel behind the lm_head has the answer
            std::vector<int32_t> am(bs);
            if (ps_hip_model_argmax(be.m_model, (int)bs, am.data())) POWERSERVE_ABORT(std::string("arg-max copy: ") + ps_hip_last_error(be.m_ctx));
            ids->assign(am.begin(), am.end());
            return LogitsVector();
        }
        Stride st = {4, 4 * (size_t)llm.vocab_size, 4 * (size_t)llm.vocab_size * bs, 4 * (size_t)llm.vocab_size * bs};
        auto host = std::make_shared<CPUBuffer>(st, (size_t)llm.vocab_size * bs * 4);
        be.sync();
        if (ps_hip_memcpy_d2h(be.m_ctx, host->m_data, logits_dev, host->m_storage.size())) POWERSERVE_ABORT(std::string("logits copy: ") + ps_hip_last_error(be.m_ctx));
        return LogitsVector(host, llm.vocab_size, bs);
    };
    const bool consecutive = [&] { for (size_t i = 1; i < bs; i++) if (pos[i] != pos[0] + (int)i) return false; return (size_t)pos[0] == m_platform->get_kv_position(m_config->model_id); }();
    if (m_use_fused && m_use_plan_cache && !plan_only && mask.mask.empty() && consecutive && m_lowered_shapes.count({bs, lm_head})) {
        m_last_lowered = true;
        n_plan_cache_hits++;
        std::vector<int32_t> t(tokens.begin(), tokens.end()), p(pos.begin(), pos.end());
        run_checked([&] {
            if (ps_hip_model_forward_lowered(be.m_model, t.data(), (int)bs, p.data(), nullptr, lm_head ? 1 : 0)) POWERSERVE_ABORT(std::string("lowered forward: ") + ps_hip_last_error(be.m_ctx));
        });
        return finish(ps_hip_model_logits(be.m_model), true);
    }
    Graph g(m_config->model_id);
    auto x = g.get_embedding(g.add_tensor(m_weights->token_embedding_table), tokens);
    TensorNode *logits = nullptr;
    be.reset_kv_batch_size(bs);
    for (size_t L = 0; L < llm.n_layers; L++) {
        auto [k_cache, v_cache] = be.m_kv->get_cache(L);
        auto att_o = m_attn->build(g, x, (int64_t)L, g.add_tensor(k_cache), g.add_tensor(v_cache), pos, mask, m_is_need_bias);
        x = m_ffn->build(g, att_o, (int64_t)L);
    }
    if (lm_head) {
        auto normed = g.rms_norm(x, g.add_tensor(m_weights->rms_final_weight), llm.norm_eps);
        logits = g.mat_mul(g.add_tensor(m_weights->output_weight), normed);
    }
    if (plan_only) { // (Model::prefill asks whether this chunk shape lowers; nothing runs, and the backend forgets the probe's lowering: be.lowered() describes graphs that ran)
        Executor executor(*m_platform, g);
        executor.plan();
        m_last_lowered = executor.lowered();
        if (m_last_lowered && m_use_plan_cache && mask.mask.empty()) m_lowered_shapes.insert({bs, lm_head});
        be.discard_plan();
        return LogitsVector();
    }
    run_checked([&] {
        Executor executor(*m_platform, g);
        executor.plan();
        m_last_lowered = executor.lowered();
        if (m_last_lowered && m_use_plan_cache && mask.mask.empty()) m_lowered_shapes.insert({bs, lm_head});
        if (!executor.lowered()) executor.allocate_buffers(); // a lowered graph runs in the device model's own arena
        executor.run(); // (a lowered single-token forward is only enqueued: run_checked looks at its attention's flag before anything counts)
    });
    return finish(logits ? logits->get<HIPBuffer>().m_data : nullptr, m_last_lowered);
}

auto Model::decode(const std::vector<Token> &tokens, const std::vector<int> &pos, bool lm_head) -> std::vector<Token> {
    std::vector<int> t(tokens.begin(), tokens.end());
    POWERSERVE_ASSERT(tokens.size() == pos.size() && !tokens.empty());
    backend().m_fused = m_use_fused;
    std::vector<Token> out;
    auto ret = forward_graph(t, pos, CausalAttentionMask(tokens.size()), lm_head, &out);
    if (out.empty())
        for (auto lg : ret.logits_vector) out.push_back((Token)(std::max_element(lg.begin(), lg.end()) - lg.begin())); // greedy_sample
    return out;
}

void Model::prefill(const std::vector<Token> &tokens, size_t batch_size) {
    if (tokens.empty()) return;
    POWERSERVE_ASSERT(batch_size > 0);
    auto &be = backend();
    auto &id = m_config->model_id;
    be.m_fused = m_use_fused;
    const size_t first = std::min(batch_size, tokens.size());
    bool lowered = false;
    if (m_use_fused && batch_size <= (size_t)ps_hip_model_max_batch(be.m_model)) { // does a chunk of this model's canonical graph lower?  (plan only: nothing runs)
        std::vector<int> t(tokens.begin(), tokens.begin() + first), pos(first);
        std::iota(pos.begin(), pos.end(), (int)m_platform->get_kv_position(id));
        forward_graph(t, pos, CausalAttentionMask(first), false, nullptr, true);
        lowered = m_last_lowered;
    }
    if (lowered) {
        std::vector<int32_t> t(tokens.begin(), tokens.end());
        if (ps_hip_model_prefill(be.m_model, t.data(), (int)t.size(), (int)batch_size)) POWERSERVE_ABORT(std::string("prefill: ") + ps_hip_last_error(be.m_ctx));
        return;
    }
    for (size_t done = 0; done < tokens.size();) {
        const size_t bs = std::min(batch_size, tokens.size() - done);
        std::vector<Token> toks(tokens.begin() + done, tokens.begin() + done + bs);
        std::vector<int> pos(bs);
        std::iota(pos.begin(), pos.end(), (int)m_platform->get_kv_position(id));
        decode(toks, pos, false);
        done += bs;
    }
}

auto Model::generate(const std::vector<Token> &prompt, int steps, size_t batch_size) -> std::vector<Token> {
    std::vector<Token> out;
    if (steps <= 0 || prompt.empty()) return out;
    auto &id = m_config->model_id;
    m_platform->reset_kv_position(id);
    backend().setup_threadpool();
    prefill(std::vector<Token>(prompt.begin(), prompt.end() - 1), batch_size); // the last prompt token is the first decode input (model.hpp:147-163)
    if (m_use_fused) { // device-side greedy loop: hipGraph replay, ids stay on the GPU until the end
        out.resize(steps);
        if (ps_hip_model_decode_greedy(backend().m_model, prompt.back(), steps, out.data()))
            POWERSERVE_ABORT(std::string("decode_greedy: ") + ps_hip_last_error(backend().m_ctx));
    } else {
        Token cur = prompt.back();
        for (int s = 0; s < steps; s++) {
            auto r = decode({cur}, {(int)m_platform->get_kv_position(id)}, true);
            cur = r[0];
            out.push_back(cur);
        }
    }
    backend().reset_threadpool();
    return out;
}

auto Model::generate(const std::vector<Token> &prompt, int steps, size_t batch_size, SamplerChain &sampler) -> std::vector<Token> {
    std::vector<Token> out;
    if (steps <= 0 || prompt.empty()) return out;
    auto &id = m_config->model_id;
    m_platform->reset_kv_position(id);
    prefill(std::vector<Token>(prompt.begin(), prompt.end() - 1), batch_size); // the last prompt token is the first decode input (model.hpp:147-163)
    Token cur = prompt.back();
    for (int s = 0; s < steps; s++) {
        auto ret = forward({cur}, {(int)m_platform->get_kv_position(id)}, CausalAttentionMask(1), true);
        cur = sampler.sample(ret.logits_vector[0]);
        out.push_back(cur);
    }
    return out;
}

auto load_model(const std::string &model_dir, const std::shared_ptr<Platform> &platform, int device, size_t max_batch, int n_ctx_cap)
    -> std::shared_ptr<Model> {
    auto config = std::make_shared<ModelConfig>(model_dir + "/model.json");
    if (n_ctx_cap > 0) config->llm.seq_len = (uint32_t)n_ctx_cap; // explicit cap: the FP32 KV cache is sized by n_ctx
    if (config->arch != "llama" && config->arch != "qwen2") POWERSERVE_ABORT("unknown model type: " + config->arch);
    return std::make_shared<Model>(model_dir, config, platform, device, max_batch);
}

} // namespace powerserve

// ---------------------------------------------------------------- C driver API (ctypes / tests / tools)
using namespace powerserve;
static thread_local std::string g_err;
extern "C" {
const char *psh_last_error(void) { return g_err.c_str(); }
void psh_set_error(const char *msg) { g_err = msg; }
void *psh_model_load(const char *model_dir, int device, int max_batch, int n_ctx_cap) {
    try {
        auto h = new psh_model();
        h->platform = std::make_shared<Platform>();
        h->model = load_model(model_dir, h->platform, device, (size_t)max_batch, n_ctx_cap);
        return h;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void psh_model_free(void *h) { delete (psh_model *)h; }
void psh_model_set_fused(void *h, int fused) { ((psh_model *)h)->model->m_use_fused = fused != 0; } // 0: plan() lowers nothing (A/B, tests)
void psh_model_plan_stats(void *h, int *n_plans, int *n_lowered) { auto &be = ((psh_model *)h)->model->backend(); *n_plans = be.n_plans; *n_lowered = be.n_lowered; }
int psh_model_plan_cache_hits(void *h) { return ((psh_model *)h)->model->n_plan_cache_hits; }
void psh_model_set_plan_cache(void *h, int on) { ((psh_model *)h)->model->m_use_plan_cache = on != 0; if (!on) ((psh_model *)h)->model->m_lowered_shapes.clear(); }
size_t psh_model_kv_position(void *h) { auto m = (psh_model *)h; return m->platform->get_kv_position(m->model->m_config->model_id); }
void psh_model_reset(void *h) { auto m = (psh_model *)h; m->platform->reset_kv_position(m->model->m_config->model_id); }
uint32_t psh_model_vocab(void *h) { return ((psh_model *)h)->model->m_config->llm.vocab_size; }
int psh_model_forward(void *h, const int32_t *tokens, int n, const int32_t *pos, int lm_head, float *logits_out) {
    try {
        auto m = (psh_model *)h;
        std::vector<int> t(tokens, tokens + n), p(pos, pos + n);
        auto r = m->model->forward(t, p, CausalAttentionMask(n), lm_head != 0);
        if (lm_head && logits_out)
            for (int i = 0; i < n; i++) memcpy(logits_out + (size_t)i * r.logits_vector[i].size(), r.logits_vector[i].data(), r.logits_vector[i].size() * 4);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
int psh_model_decode(void *h, const int32_t *tokens, int n, const int32_t *pos, int32_t *ids_out) { // Model::decode, greedy: ids only
    try {
        auto m = (psh_model *)h;
        std::vector<Token> t(tokens, tokens + n);
        std::vector<int> p(pos, pos + n);
        auto r = m->model->decode(t, p, true);
        for (int i = 0; i < n; i++) ids_out[i] = r[i];
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
int psh_model_prefill(void *h, const int32_t *tokens, int n, int batch_size) { // ModelTokenIterator's prefill loop
    try {
        auto m = (psh_model *)h;
        m->model->prefill(std::vector<Token>(tokens, tokens + n), (size_t)batch_size);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
// ---- boundary members no model graph uses (tests call each of them once: Graph::softmax through the executor, get_n_tasks, add_cache,
// and the KVCacheInterface members of src/core/kv_cache.hpp:120-162 on the device cache)
int psh_graph_softmax(void *h, const float *x_host, int64_t n, int64_t rows, float *out_host) { // SOFTMAX op: Graph::softmax -> Executor -> HIPBackend::softmax -> ps_hip_soft_max
    try {
        auto m = (psh_model *)h;
        auto &be = m->model->backend();
        Graph g(m->model->m_config->model_id);
        auto x = g.new_tensor(DataType::FP32, {(size_t)n, (size_t)rows, 1, 1});
        auto y = g.softmax(x);
        Executor ex(*m->platform, g);
        ex.plan();
        POWERSERVE_ASSERT(!ex.lowered());
        ex.allocate_buffers();
        if (ps_hip_memcpy_h2d(be.m_ctx, x->get<HIPBuffer>().m_data, x_host, (size_t)n * rows * 4)) POWERSERVE_ABORT(ps_hip_last_error(be.m_ctx));
        ex.run();
        be.sync();
        if (ps_hip_memcpy_d2h(be.m_ctx, out_host, y->get<HIPBuffer>().m_data, (size_t)n * rows * 4)) POWERSERVE_ABORT(ps_hip_last_error(be.m_ctx));
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
int psh_backend_get_n_tasks(void *h) {
    auto op = std::make_shared<OpNode>(OpType::MAT_MUL);
    return ((psh_model *)h)->model->backend().get_n_tasks(op);
}
int psh_backend_add_cache(void *h, int L, const float *k_host, const float *v_host, int bs) { // k, v: [bs][kv_dim] rows of the batch
    try {
        auto m = (psh_model *)h;
        auto &be = m->model->backend();
        const size_t kvd = be.m_kv->m_kv_dim, bytes = kvd * (size_t)bs * 4;
        be.reset_kv_batch_size((size_t)bs);
        be.arena_reset();
        be.arena_reserve(2 * bytes + 4096);
        Tensor k(DataType::FP32, {kvd, (size_t)bs, 1, 1}), v(DataType::FP32, {kvd, (size_t)bs, 1, 1});
        const Stride st{4, 4 * kvd, bytes, bytes};
        k.m_data = std::make_shared<HIPBuffer>(st, be.arena_alloc(bytes));
        v.m_data = std::make_shared<HIPBuffer>(st, be.arena_alloc(bytes));
        if (ps_hip_memcpy_h2d(be.m_ctx, k.get<HIPBuffer>().m_data, k_host, bytes) || ps_hip_memcpy_h2d(be.m_ctx, v.get<HIPBuffer>().m_data, v_host, bytes))
            POWERSERVE_ABORT(ps_hip_last_error(be.m_ctx));
        std::vector<int> pos(bs);
        std::iota(pos.begin(), pos.end(), (int)be.m_kv->position());
        be.add_cache(&k, &v, (size_t)L, pos, 0);
        be.sync();
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
int psh_model_kv_read(void *h, int L, int64_t slot, float *k_out, float *v_out) { // cache slot `slot` of layer L: its K row and its V column, kv_dim floats each
    try {
        auto &be = ((psh_model *)h)->model->backend();
        const size_t kvd = be.m_kv->m_kv_dim, n_ctx = be.m_kv->m_n_ctx;
        be.sync();
        std::vector<float> vt(kvd * n_ctx);
        if (ps_hip_memcpy_d2h(be.m_ctx, k_out, ps_hip_model_k_cache(be.m_model, L) + (size_t)slot * kvd, kvd * 4) ||
            ps_hip_memcpy_d2h(be.m_ctx, vt.data(), ps_hip_model_v_cache(be.m_model, L), vt.size() * 4))
            POWERSERVE_ABORT(ps_hip_last_error(be.m_ctx));
        for (size_t d = 0; d < kvd; d++) v_out[d] = vt[d * n_ctx + (size_t)slot];
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
// op: 0 copy(a, b) 1 move(a, b) 2 mask(a) 3 unmask(a) 4 save_tokens(a) 5 unmask_tokens(a) 6 advance_tokens(a) 7 rollback_tokens(a) 8 truncate_tokens(a) 9 append_tokens(a);
// returns what the member returns (the old position for 6..9, else 0), -1 on failure
int64_t psh_kv_op(void *h, int op, int64_t a, int64_t b) {
    try {
        auto &kv = *((psh_model *)h)->model->backend().m_kv;
        switch (op) {
        case 0: kv.copy((size_t)a, (size_t)b); return 0;
        case 1: kv.move((size_t)a, (size_t)b); return 0;
        case 2: kv.mask((size_t)a); return 0;
        case 3: kv.unmask((size_t)a); return 0;
        case 4: kv.save_tokens((size_t)a); return 0;
        case 5: kv.unmask_tokens((size_t)a); return 0;
        case 6: return (int64_t)kv.advance_tokens((size_t)a);
        case 7: return (int64_t)kv.rollback_tokens((size_t)a);
        case 8: return (int64_t)kv.truncate_tokens((size_t)a);
        case 9: return (int64_t)kv.append_tokens((size_t)a);
        }
        g_err = "psh_kv_op: unknown op";
        return -1;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// sampler chain alone (tests against the reference's samplers) and sampled generation
struct psh_sampler_cfg { // plain-C view of SamplerConfig + the two vocabulary ids the chain needs
    uint64_t seed;
    float temperature, top_p;
    uint64_t top_k;
    int32_t penalty_last_n;
    float penalty_repeat, penalty_freq, penalty_present;
    int32_t penalize_nl, ignore_eos, n_vocabs, special_eos_id, linefeed_id;
};
static SamplerChain *make_chain(const psh_sampler_cfg *c) {
    SamplerConfig sc;
    sc.seed = c->seed; sc.temperature = c->temperature; sc.top_p = c->top_p; sc.top_k = (size_t)c->top_k;
    sc.penalty_last_n = c->penalty_last_n; sc.penalty_repeat = c->penalty_repeat; sc.penalty_freq = c->penalty_freq;
    sc.penalty_present = c->penalty_present; sc.penalize_nl = c->penalize_nl != 0; sc.ignore_eos = c->ignore_eos != 0;
    return new SamplerChain(sc, c->n_vocabs, c->special_eos_id, c->linefeed_id);
}
void *psh_sampler_create(const psh_sampler_cfg *c) {
    try { return make_chain(c); } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void psh_sampler_free(void *s) { delete (SamplerChain *)s; }
int32_t psh_sampler_sample(void *s, const float *logits, int n) {
    try { return ((SamplerChain *)s)->sample(std::span<const float>(logits, (size_t)n)); } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
int psh_model_generate_sampled(void *h, const int32_t *prompt, int n_prompt, int batch_size, int steps, const psh_sampler_cfg *c, int32_t *out) {
    try {
        auto m = (psh_model *)h;
        std::unique_ptr<SamplerChain> chain(make_chain(c));
        std::vector<Token> p(prompt, prompt + n_prompt);
        auto r = m->model->generate(p, steps, (size_t)batch_size, *chain);
        memcpy(out, r.data(), r.size() * 4);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
int psh_model_generate(void *h, const int32_t *prompt, int n_prompt, int batch_size, int steps, int32_t *out) {
    try {
        auto m = (psh_model *)h;
        std::vector<Token> p(prompt, prompt + n_prompt);
        auto r = m->model->generate(p, steps, (size_t)batch_size);
        memcpy(out, r.data(), r.size() * 4);
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 1; }
}
}
