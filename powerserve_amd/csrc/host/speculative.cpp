#include "speculative.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace powerserve {

// ---------------------------------------------------------------- the tree's view of a model, on the HIP backend
void HIPSpecBackend::check(int rc, const char *what) {
    if (rc) POWERSERVE_ABORT(std::string(what) + ": " + ps_hip_last_error(m.backend().m_ctx));
}
size_t HIPSpecBackend::kv_position() { return ps_hip_model_kv_position(m.backend().m_model); }
void HIPSpecBackend::forward_one(Token token, int position, std::vector<float> *logits) {
    int32_t t = token, p = position, am = 0;
    check(ps_hip_model_forward_tree(m.backend().m_model, &t, 1, &p, nullptr, logits ? 1 : 0, logits ? &am : nullptr, 1), "draft forward");
    if (logits) {
        logits->resize(m.m_config->llm.vocab_size);
        check(ps_hip_memcpy_d2h(m.backend().m_ctx, logits->data(), ps_hip_model_logits(m.backend().m_model), logits->size() * 4), "logits copy");
    }
}
void HIPSpecBackend::forward_tree(const std::vector<int32_t> &tokens, const std::vector<int32_t> &positions, const std::vector<uint8_t> &mask,
                                  std::vector<int32_t> &argmax) {
    argmax.resize(tokens.size());
    // advance = 0: the reference's forward followed by rollback_tokens(draft_batch_size) (spec_model.hpp:98-102)
    check(ps_hip_model_forward_tree(m.backend().m_model, tokens.data(), (int)tokens.size(), positions.data(), mask.data(), 1, argmax.data(), 0), "tree verify");
}
bool HIPSpecBackend::tree_logits(size_t n, std::vector<float> &logits) {
    logits.resize(n * vocab_size());
    check(ps_hip_memcpy_d2h(m.backend().m_ctx, logits.data(), ps_hip_model_logits(m.backend().m_model), logits.size() * 4), "tree logits copy");
    return true;
}
size_t HIPSpecBackend::vocab_size() const { return m.m_config->llm.vocab_size; }
size_t HIPSpecBackend::n_ctx() const { return m.m_config->llm.seq_len; }
void HIPSpecBackend::kv_mask(size_t slot, bool visible) { check(ps_hip_model_kv_mask(m.backend().m_model, slot, visible ? 1 : 0), "kv mask"); }
void HIPSpecBackend::kv_move(size_t dst, size_t src) { check(ps_hip_model_kv_move(m.backend().m_model, dst, src), "kv move"); }
void HIPSpecBackend::kv_advance(size_t n) { check(ps_hip_model_kv_advance(m.backend().m_model, n), "kv advance"); }
void HIPSpecBackend::kv_rollback(size_t n) { check(ps_hip_model_kv_rollback(m.backend().m_model, n), "kv rollback"); }

// ---------------------------------------------------------------- ... and on caller-supplied functions
static void cb_check(int rc, const char *what) {
    if (rc) POWERSERVE_ABORT(std::string("speculative backend callback failed: ") + what);
}
size_t CallbackSpecBackend::kv_position() {
    const int64_t p = cb.kv_position(cb.user);
    if (p < 0) POWERSERVE_ABORT("speculative backend callback failed: kv_position");
    return (size_t)p;
}
void CallbackSpecBackend::forward_one(Token token, int position, std::vector<float> *logits) {
    if (logits) logits->resize((size_t)cb.vocab_size);
    cb_check(cb.forward_one(cb.user, token, position, logits ? logits->data() : nullptr), "forward_one");
}
void CallbackSpecBackend::forward_tree(const std::vector<int32_t> &tokens, const std::vector<int32_t> &positions, const std::vector<uint8_t> &mask,
                                       std::vector<int32_t> &argmax) {
    argmax.resize(tokens.size());
    cb_check(cb.forward_tree(cb.user, tokens.data(), (int32_t)tokens.size(), positions.data(), mask.data(), argmax.data()), "forward_tree");
}
void CallbackSpecBackend::kv_mask(size_t slot, bool visible) { cb_check(cb.kv_mask(cb.user, (int64_t)slot, visible ? 1 : 0), "kv_mask"); }
void CallbackSpecBackend::kv_move(size_t dst, size_t src) { cb_check(cb.kv_move(cb.user, (int64_t)dst, (int64_t)src), "kv_move"); }
void CallbackSpecBackend::kv_advance(size_t n) { cb_check(cb.kv_advance(cb.user, (int64_t)n), "kv_advance"); }
void CallbackSpecBackend::kv_rollback(size_t n) { cb_check(cb.kv_rollback(cb.user, (int64_t)n), "kv_rollback"); }

// ---------------------------------------------------------------- token tree
std::vector<ProbIndex> draft_sample(std::span<const float> logits, size_t top_k, float temperature) {
    ProbArray c(logits);
    stage::top_k(c, top_k);
    stage::temperature(c, temperature);
    stage::softmax(c);
    return std::move(c.m_probs);
}

namespace {
// A continuation waiting to become a node.  Ordered by the probability of the whole path from the root ONLY (as
// token_tree.hpp:73-81): which of two equally likely candidates surfaces first is then decided by the heap algorithm
// of std::priority_queue, the same container the reference keeps them in.
struct Pending {
    Token token;
    int parent;
    float prob, path_prob;
    bool operator<(const Pending &o) const { return path_prob < o.path_prob; }
};
} // namespace

std::vector<int32_t> TokenTree::tokens() const {
    std::vector<int32_t> t;
    for (const Node &n : m_nodes) t.push_back(n.token);
    return t;
}
std::vector<int32_t> TokenTree::positions() const {
    std::vector<int32_t> p;
    for (const Node &n : m_nodes) p.push_back(n.position);
    return p;
}
std::vector<uint8_t> TokenTree::attention_mask() const {
    const size_t n = m_nodes.size();
    std::vector<uint8_t> visible(n * n, 0);
    for (size_t u = 0; u < n; u++)
        for (int a = (int)u; a != NO_PARENT; a = m_nodes[a].parent) visible[u * n + (size_t)a] = 1;
    return visible;
}

int TokenTree::lca(int u, int v) const {
    while (u != v) { // lift the deeper one (both when level)
        const int du = m_nodes[u].depth, dv = m_nodes[v].depth;
        if (du >= dv) u = m_nodes[u].parent;
        if (dv >= du) v = m_nodes[v].parent;
    }
    return u;
}

// The draft model's next forward must see the path root..new_parent and nothing else of the tree: hide the slots of the
// old branch below the common ancestor (walking up from old_parent), then show the new branch's (up from new_parent).
void TokenTree::switch_parent(SpecBackend &draft_model, int old_parent, int new_parent) {
    if (old_parent == new_parent) return;
    const int fork = lca(old_parent, new_parent);
    for (int x = old_parent; x != fork; x = m_nodes[x].parent) {
        POWERSERVE_ASSERT(m_nodes[x].cache_index != NOT_IN_CACHE);
        draft_model.kv_mask((size_t)m_nodes[x].cache_index, false);
    }
    for (int x = new_parent; x != fork; x = m_nodes[x].parent) {
        POWERSERVE_ASSERT(m_nodes[x].cache_index != NOT_IN_CACHE);
        draft_model.kv_mask((size_t)m_nodes[x].cache_index, true);
    }
}

void TokenTree::draft(SpecBackend &draft_model, size_t batch_size, Token root_token, const std::function<bool(Token)> &should_stop) {
    const auto &shape = m_config.token_tree;
    const auto &smp   = m_config.draft_sampler;
    m_nodes.clear();
    m_nodes.reserve(batch_size);
    // `open`: candidates that may be expanded with a draft forward; `closed`: ones that can only fill leftover slots
    std::priority_queue<Pending> open, closed;
    open.push({root_token, NO_PARENT, 1.0f, 1.0f});
    int expanded_last = NO_PARENT;
    size_t n_forwards = 0;
    std::vector<float> logits;
    while (m_nodes.size() < batch_size) {
        const bool filling = open.empty();
        auto &from = filling ? closed : open;
        if (from.empty()) break;
        const Pending c = from.top();
        from.pop();

        const int u = (int)m_nodes.size();
        Node &node = m_nodes.emplace_back();
        node.token = c.token;
        node.current_prob = c.prob;
        if (c.parent == NO_PARENT) {
            node.position = (int)draft_model.kv_position();
        } else {
            Node &up = m_nodes[c.parent];
            node.parent = c.parent;
            node.depth = up.depth + 1;
            node.position = up.position + 1;
            up.children.push_back(u);
        }

        // a node is left unexpanded when it came from the closed heap, ends the text, is too unlikely, or when the slots
        // that remain are spoken for (with early_stop, half of the open candidates count as spoken for)
        const size_t reserved = shape.early_stop ? open.size() / 2 : 0;
        if (filling || (should_stop && should_stop(c.token)) || m_nodes.size() + reserved >= batch_size || c.path_prob < shape.min_prob) continue;

        if (expanded_last != NO_PARENT) switch_parent(draft_model, expanded_last, c.parent);
        node.cache_index = (int)draft_model.kv_position();
        draft_model.forward_one(c.token, node.position, &logits);
        n_forwards++;
        expanded_last = u;

        const auto next = draft_sample(logits, smp.top_k, smp.temperature);
        POWERSERVE_ASSERT(!next.empty(), "the draft sampler returned no candidate (top_k = 0 or an empty vocabulary)");
        const float floor = next[0].prob * smp.p_base;
        for (size_t i = 0; i < next.size(); i++) {
            const bool fill_only = i >= shape.max_fan_out || next[i].prob < floor;
            (fill_only ? closed : open).push({next[i].token, u, next[i].prob, c.path_prob * next[i].prob});
        }
    }
    // (the reference keeps batch_size nodes, the unused ones blank, because its NPU graphs have a fixed batch shape;
    //  here the verify batch is just the nodes that exist)
    m_stat.n_draft_times += n_forwards;
    m_stat.n_draft_tokens += m_nodes.size() - 1;
    draft_model.kv_rollback(n_forwards);
}

void TokenTree::verify(SpecBackend &target_model, SpecBackend &draft_model, const std::function<Token(int)> &choose, const std::function<void(Token)> &enqueue) {
    POWERSERVE_ASSERT(target_model.kv_position() == draft_model.kv_position());
    m_stat.n_iterations++;
    const size_t staging = target_model.kv_position(); // the tree forward left node u's KV in target slot staging + u
    for (int u = 0;;) {
        Node &node = m_nodes[u];
        node.accepted = true;
        POWERSERVE_ASSERT((int)draft_model.kv_position() == node.position && (int)target_model.kv_position() == node.position);
        // target: keep this node's KV (src >= dst always: a node at depth d has index >= d)
        target_model.kv_move((size_t)node.position, staging + (size_t)u);
        target_model.kv_advance(1);
        // draft: reuse the KV from drafting if the node was expanded, else evaluate it now
        if (node.cache_index == NOT_IN_CACHE) {
            draft_model.forward_one(node.token, node.position, nullptr);
        } else {
            POWERSERVE_ASSERT(node.cache_index >= node.position);
            draft_model.kv_move((size_t)node.position, (size_t)node.cache_index);
            draft_model.kv_advance(1);
        }
        const Token chosen = choose(u);
        enqueue(chosen);
        m_stat.n_generated_tokens++;
        const auto child = std::find_if(node.children.begin(), node.children.end(), [&](int v) { return m_nodes[v].token == chosen; });
        if (child == node.children.end()) break;
        u = *child;
        m_stat.n_accepted_tokens++;
    }
}

void TokenTree::iterate(SpecBackend &target_model, SpecBackend &draft_model, Token last, std::vector<Token> &out, Sampler *sampler,
                        const std::function<bool(Token)> &should_stop) {
    draft(draft_model, m_config.draft_batch_size, last, should_stop);
    std::vector<int32_t> argmax;
    target_model.forward_tree(tokens(), positions(), attention_mask(), argmax);
    if (!sampler) {
        verify(target_model, draft_model, argmax, [&](Token t) { out.push_back(t); });
        return;
    }
    std::vector<float> logits;
    const size_t V = target_model.vocab_size();
    if (V == 0 || !target_model.tree_logits(m_nodes.size(), logits)) POWERSERVE_ABORT("speculative verify with a sampler needs a target backend that returns the tree's logits");
    verify(target_model, draft_model, [&](int u) {
        ProbArray probs(std::span<const float>(logits.data() + (size_t)u * V, V)); // token_tree.cpp:214-216
        sampler->apply(probs);
        return probs.greedy_sample().token;
    }, [&](Token t) { out.push_back(t); });
}

// ModelTokenIterator's prefill (model.hpp:117-150): everything but the last prompt token, no lm_head
static void prefill(Model &m, const std::vector<Token> &prompt, size_t batch_size) {
    if (ps_hip_model_kv_truncate(m.backend().m_model, 0)) POWERSERVE_ABORT(std::string("kv truncate: ") + ps_hip_last_error(m.backend().m_ctx));
    for (size_t done = 0; done + 1 < prompt.size();) {
        const size_t bs = std::min(batch_size, prompt.size() - 1 - done);
        std::vector<int> toks(prompt.begin() + done, prompt.begin() + done + bs), pos(bs);
        std::iota(pos.begin(), pos.end(), (int)done);
        m.forward(toks, pos, CausalAttentionMask(bs), false);
        done += bs;
    }
}

std::vector<Token> SpeculativeModel::generate(const std::vector<Token> &prompt, int steps, size_t batch_size, Sampler *sampler,
                                              const std::function<bool(Token)> &should_stop) {
    std::vector<Token> out;
    if (steps <= 0 || prompt.empty()) return out;
    prefill(*target_model, prompt, batch_size);
    prefill(*draft_model, prompt, batch_size);
    HIPSpecBackend target(*target_model), draft(*draft_model);
    Token last = prompt.back();
    bool stopped = false;
    while ((int)out.size() < steps && !stopped) {
        // a tree needs draft_batch_size free slots behind the position in both caches: end the text instead of aborting
        if (target.kv_position() + config.draft_batch_size > target.n_ctx() || draft.kv_position() + config.draft_batch_size > draft.n_ctx()) break;
        const size_t from = out.size();
        token_tree.iterate(target, draft, last, out, sampler, should_stop);
        // No sampler->accept() here: in the reference only Model::decode accepts (llama_model.cpp:128); SpecTokenIterator::decode and
        // TokenTree::verify never do, so the repeat-penalty history keeps its initial window for the whole speculative text
        // (tests/test_gpu_speculative.py::test_speculative_sampler_history_is_never_advanced pins this).
        for (size_t i = from; i < out.size(); i++)
            if (should_stop && should_stop(out[i])) { out.resize(i + 1); stopped = true; break; }
        last = out.back();
    }
    if ((int)out.size() > steps) out.resize((size_t)steps);
    return out;
}

} // namespace powerserve

// ---------------------------------------------------------------- C driver API
using namespace powerserve;
extern "C" {
extern const char *psh_last_error(void);
void psh_set_error(const char *msg);

typedef struct psh_spec_config { // plain-C view of SpeculativeConfig
    int32_t draft_batch_size, top_k, max_fan_out, early_stop;
    float temperature, p_base, min_prob;
} psh_spec_config;

static SpeculativeConfig make_config(const psh_spec_config *c) {
    SpeculativeConfig cfg;
    if (!c) return cfg;
    if (c->draft_batch_size > 0) cfg.draft_batch_size = (size_t)c->draft_batch_size;
    if (c->top_k > 0) cfg.draft_sampler.top_k = (size_t)c->top_k;
    if (c->temperature > 0) cfg.draft_sampler.temperature = c->temperature;
    if (c->p_base > 0) cfg.draft_sampler.p_base = c->p_base;
    if (c->max_fan_out > 0) cfg.token_tree.max_fan_out = (size_t)c->max_fan_out;
    if (c->min_prob >= 0) cfg.token_tree.min_prob = c->min_prob;
    if (c->early_stop >= 0) cfg.token_tree.early_stop = c->early_stop != 0;
    return cfg;
}

// speculative generate over two loaded models; stats: {n_draft_times, n_draft_tokens, n_accepted_tokens, n_iterations, n_generated_tokens}
int psh_spec_generate(void *target, void *draft, const int32_t *prompt, int n_prompt, int batch_size, int steps, int draft_batch_size,
                      int32_t *out, uint64_t *stats) {
    try {
        SpeculativeConfig cfg;
        if (draft_batch_size > 0) cfg.draft_batch_size = (size_t)draft_batch_size;
        SpeculativeModel sm(((psh_model *)target)->model, ((psh_model *)draft)->model, cfg);
        std::vector<Token> p(prompt, prompt + n_prompt);
        const auto r = sm.generate(p, steps, (size_t)batch_size);
        memcpy(out, r.data(), r.size() * 4);
        if (stats) {
            const auto &s = sm.token_tree.m_stat;
            stats[0] = s.n_draft_times; stats[1] = s.n_draft_tokens; stats[2] = s.n_accepted_tokens; stats[3] = s.n_iterations; stats[4] = s.n_generated_tokens;
        }
        return 0;
    } catch (const std::exception &e) { psh_set_error(e.what()); return 1; }
}

// the same with the verify going through a sampler chain (handle of psh_sampler_create; spec_model.hpp:105) and an optional
// stop token (eos < 0: none).  The text ends early at the stop token or when the caches cannot hold another tree: *n_out <= steps.
int psh_spec_generate_sampled(void *target, void *draft, const int32_t *prompt, int n_prompt, int batch_size, int steps, int draft_batch_size,
                              void *sampler, int32_t eos, int32_t *out, int32_t *n_out, uint64_t *stats) {
    try {
        SpeculativeConfig cfg;
        if (draft_batch_size > 0) cfg.draft_batch_size = (size_t)draft_batch_size;
        SpeculativeModel sm(((psh_model *)target)->model, ((psh_model *)draft)->model, cfg);
        std::vector<Token> p(prompt, prompt + n_prompt);
        std::function<bool(Token)> stop;
        if (eos >= 0) stop = [eos](Token t) { return t == eos; };
        const auto r = sm.generate(p, steps, (size_t)batch_size, (SamplerChain *)sampler, stop);
        if (!r.empty()) memcpy(out, r.data(), r.size() * 4);
        *n_out = (int32_t)r.size();
        if (stats) {
            const auto &s = sm.token_tree.m_stat;
            stats[0] = s.n_draft_times; stats[1] = s.n_draft_tokens; stats[2] = s.n_accepted_tokens; stats[3] = s.n_iterations; stats[4] = s.n_generated_tokens;
        }
        return 0;
    } catch (const std::exception &e) { psh_set_error(e.what()); return 1; }
}

// The token tree over caller-supplied models: n_iterations rounds of draft / tree forward / verify starting from
// root_token (both caches already hold the same prefix).  out_tokens: capacity n_iterations * draft_batch_size,
// *n_out = tokens emitted.  tree (optional): per iteration draft_batch_size rows of
// {token, position, parent, cache_index, accepted, depth}, n_nodes[it] rows used.
int psh_token_tree_run(const psh_spec_backend *target, const psh_spec_backend *draft, const psh_spec_config *config, int32_t root_token,
                       int n_iterations, int32_t *out_tokens, int32_t *n_out, int32_t *tree, int32_t *n_nodes, uint64_t *stats) {
    try {
        const SpeculativeConfig cfg = make_config(config);
        CallbackSpecBackend t(*target), d(*draft);
        TokenTree tt(cfg);
        std::vector<Token> out;
        Token last = root_token;
        for (int it = 0; it < n_iterations; it++) {
            tt.iterate(t, d, last, out);
            last = out.back();
            if (n_nodes) n_nodes[it] = (int32_t)tt.m_nodes.size();
            if (tree) {
                int32_t *row = tree + (size_t)it * cfg.draft_batch_size * 6;
                for (const auto &n : tt.m_nodes) {
                    row[0] = n.token; row[1] = n.position; row[2] = n.parent; row[3] = n.cache_index; row[4] = n.accepted; row[5] = n.depth;
                    row += 6;
                }
            }
        }
        memcpy(out_tokens, out.data(), out.size() * 4);
        *n_out = (int32_t)out.size();
        if (stats) {
            const auto &s = tt.m_stat;
            stats[0] = s.n_draft_times; stats[1] = s.n_draft_tokens; stats[2] = s.n_accepted_tokens; stats[3] = s.n_iterations; stats[4] = s.n_generated_tokens;
        }
        return 0;
    } catch (const std::exception &e) { psh_set_error(e.what()); return 1; }
}

// draft sampler alone (TopK -> Temperature -> Softmax): returns the number of entries written
int psh_draft_sample(const float *logits, int n, int top_k, float temperature, int32_t *tokens, float *probs) {
    try {
        const auto r = draft_sample(std::span<const float>(logits, (size_t)n), (size_t)top_k, temperature);
        for (size_t i = 0; i < r.size(); i++) { tokens[i] = r[i].token; probs[i] = r[i].prob; }
        return (int)r.size();
    } catch (const std::exception &e) { psh_set_error(e.what()); return -1; }
}
}
