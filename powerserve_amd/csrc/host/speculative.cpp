#include "speculative.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace powerserve {

namespace {

ps_hip_model *dev(Model &m) { return m.backend().m_model; }
void check(Model &m, int rc, const char *what) {
    if (rc) POWERSERVE_ABORT(std::string(what) + ": " + ps_hip_last_error(m.backend().m_ctx));
}
size_t kv_position(Model &m) { return ps_hip_model_kv_position(dev(m)); }

// one token at the current cache slot with an explicit RoPE position; optionally returns its logits row
void forward_one(Model &m, Token token, int position, std::vector<float> *logits) {
    int32_t t = token, p = position, am = 0;
    check(m, ps_hip_model_forward_tree(dev(m), &t, 1, &p, nullptr, logits ? 1 : 0, logits ? &am : nullptr, 1), "draft forward");
    if (logits) {
        logits->resize(m.m_config->llm.vocab_size);
        check(m, ps_hip_memcpy_d2h(m.backend().m_ctx, logits->data(), ps_hip_model_logits(dev(m)), logits->size() * 4), "logits copy");
    }
}

void prefill(Model &m, const std::vector<Token> &prompt, size_t batch_size) { // ModelTokenIterator's prefill (model.hpp:117-150)
    check(m, ps_hip_model_kv_truncate(dev(m), 0), "kv truncate");
    size_t done = 0;
    while (done + 1 < prompt.size()) {
        const size_t bs = std::min(batch_size, prompt.size() - 1 - done);
        std::vector<int> toks(prompt.begin() + done, prompt.begin() + done + bs), pos(bs);
        std::iota(pos.begin(), pos.end(), (int)done);
        m.forward(toks, pos, CausalAttentionMask(bs), false);
        done += bs;
    }
}

struct Candidate {
    float cumulative_prob;
    size_t seq; // insertion order: deterministic tie-break
    Token token;
    int parent;
    float current_prob;
    bool operator<(const Candidate &o) const { return cumulative_prob != o.cumulative_prob ? cumulative_prob < o.cumulative_prob : seq > o.seq; }
};

} // namespace

std::vector<ProbIndex> draft_sample(std::span<const float> logits, size_t top_k, float temperature) {
    ProbArray probs(logits); // the draft sampler chain of token_tree.cpp:35-39
    TopKSampler(top_k).apply(probs);
    TemperatureSampler(temperature).apply(probs);
    SoftmaxSampler().apply(probs);
    return probs.m_probs;
}

std::vector<int32_t> TokenTree::tokens() const {
    std::vector<int32_t> t(m_nodes.size());
    for (size_t i = 0; i < m_nodes.size(); i++) t[i] = m_nodes[i].token;
    return t;
}
std::vector<int32_t> TokenTree::positions() const {
    std::vector<int32_t> p(m_nodes.size());
    for (size_t i = 0; i < m_nodes.size(); i++) p[i] = m_nodes[i].position;
    return p;
}
std::vector<uint8_t> TokenTree::attention_mask() const {
    const size_t n = m_nodes.size();
    std::vector<uint8_t> mask(n * n, 0);
    for (size_t u = 0; u < n; u++)
        for (int x = (int)u; x != NO_PARENT; x = m_nodes[x].parent) mask[u * n + x] = 1;
    return mask;
}

int TokenTree::lca(int u, int v) const {
    if (m_nodes[u].depth < m_nodes[v].depth) std::swap(u, v);
    while (m_nodes[u].depth > m_nodes[v].depth) u = m_nodes[u].parent;
    while (u != v) { u = m_nodes[u].parent; v = m_nodes[v].parent; }
    return u;
}

// the draft model's next forward must see the path root..new_parent only: hide the old branch, show the new one
void TokenTree::switch_parent(Model &draft_model, int old_parent, int new_parent) {
    if (old_parent == new_parent) return;
    const int p = lca(old_parent, new_parent);
    for (; old_parent != p; old_parent = m_nodes[old_parent].parent)
        check(draft_model, ps_hip_model_kv_mask(dev(draft_model), (size_t)m_nodes[old_parent].cache_index, 0), "kv mask");
    for (; new_parent != p; new_parent = m_nodes[new_parent].parent)
        check(draft_model, ps_hip_model_kv_mask(dev(draft_model), (size_t)m_nodes[new_parent].cache_index, 1), "kv mask");
}

void TokenTree::draft(Model &draft_model, size_t batch_size, Token root_token, const std::function<bool(Token)> &should_stop) {
    const auto &tc = m_config.token_tree;
    const auto &sc = m_config.draft_sampler;
    m_nodes.assign(batch_size, Node());
    std::priority_queue<Candidate> main_heap, leaf_heap; // expandable candidates / candidates that may only become leaves
    size_t seq = 0;
    main_heap.push({1.0f, seq, root_token, NO_PARENT, 1.0f});
    int last_parent = NO_PARENT;
    size_t n_nodes = 0, n_saved = 0;
    std::vector<float> logits;
    while (n_nodes < batch_size) {
        const bool is_leaf = main_heap.empty();
        auto &heap = is_leaf ? leaf_heap : main_heap;
        if (heap.empty()) break;
        const Candidate c = heap.top();
        heap.pop();
        const int u = (int)n_nodes++;
        Node &node = m_nodes[u];
        node.token = c.token;
        node.current_prob = c.current_prob;
        if (c.parent == NO_PARENT) {
            node.position = (int)kv_position(draft_model);
        } else {
            node.position = m_nodes[c.parent].position + 1;
            node.parent = c.parent;
            node.depth = m_nodes[c.parent].depth + 1;
            m_nodes[c.parent].children.push_back(u);
        }
        // not expanded: leaves, stop tokens, no room left for its children, or too unlikely
        if (is_leaf || (should_stop && should_stop(c.token)) ||
            n_nodes + (tc.early_stop ? main_heap.size() / 2 : 0) >= batch_size || c.cumulative_prob < tc.min_prob)
            continue;
        if (last_parent != NO_PARENT) switch_parent(draft_model, last_parent, c.parent);
        node.cache_index = (int)kv_position(draft_model);
        forward_one(draft_model, c.token, node.position, &logits);
        n_saved++;
        last_parent = u;
        const auto probs = draft_sample(logits, sc.top_k, sc.temperature);
        const float min_prob = probs[0].prob * sc.p_base;
        for (size_t i = 0; i < probs.size(); i++) {
            const bool leaf_only = i >= tc.max_fan_out || probs[i].prob < min_prob;
            (leaf_only ? leaf_heap : main_heap).push({c.cumulative_prob * probs[i].prob, ++seq, probs[i].token, u, probs[i].prob});
        }
    }
    m_nodes.resize(n_nodes); // (the reference keeps unused default nodes for QNN's fixed batch shape; here the verify batch shrinks)
    m_stat.n_draft_times += n_saved;
    m_stat.n_draft_tokens += n_nodes - 1;
    check(draft_model, ps_hip_model_kv_rollback(dev(draft_model), n_saved), "kv rollback"); // (rollback also un-hides the slots)
}

void TokenTree::verify(Model &target_model, Model &draft_model, const std::vector<int32_t> &target_argmax, const std::function<void(Token)> &enqueue) {
    POWERSERVE_ASSERT(kv_position(target_model) == kv_position(draft_model));
    m_stat.n_iterations++;
    const size_t base = kv_position(target_model); // the tree's KV sits at cache slots base + u
    int u = 0;
    size_t n_generated = 0;
    while (true) {
        Node &node = m_nodes[u];
        node.accepted = true;
        POWERSERVE_ASSERT((int)kv_position(draft_model) == node.position && (int)kv_position(target_model) == node.position);
        check(target_model, ps_hip_model_kv_move(dev(target_model), (size_t)node.position, base + (size_t)u), "kv move");
        check(target_model, ps_hip_model_kv_advance(dev(target_model), 1), "kv advance");
        if (node.cache_index == NOT_IN_CACHE) { // the draft model never evaluated this node: catch up
            forward_one(draft_model, node.token, node.position, nullptr);
        } else {
            POWERSERVE_ASSERT(node.cache_index >= node.position);
            check(draft_model, ps_hip_model_kv_move(dev(draft_model), (size_t)node.position, (size_t)node.cache_index), "kv move");
            check(draft_model, ps_hip_model_kv_advance(dev(draft_model), 1), "kv advance");
        }
        const Token next_token = target_argmax[u];
        enqueue(next_token);
        n_generated++;
        int next = NO_PARENT;
        for (int v : node.children)
            if (m_nodes[v].token == next_token) { next = v; break; }
        if (next == NO_PARENT) break;
        u = next;
        m_stat.n_accepted_tokens++;
    }
    m_stat.n_generated_tokens += n_generated;
}

std::vector<Token> SpeculativeModel::generate(const std::vector<Token> &prompt, int steps, size_t batch_size) {
    std::vector<Token> out;
    if (steps <= 0 || prompt.empty()) return out;
    prefill(*target_model, prompt, batch_size);
    prefill(*draft_model, prompt, batch_size);
    Token last = prompt.back();
    while ((int)out.size() < steps) {
        token_tree.draft(*draft_model, config.draft_batch_size, last);
        const auto toks = token_tree.tokens(), pos = token_tree.positions();
        const auto mask = token_tree.attention_mask();
        std::vector<int32_t> am(toks.size());
        // one batched forward over the tree; advance = 0 is the reference's forward + rollback_tokens(draft_batch_size)
        check(*target_model, ps_hip_model_forward_tree(dev(*target_model), toks.data(), (int)toks.size(), pos.data(), mask.data(), 1, am.data(), 0),
              "tree verify");
        token_tree.verify(*target_model, *draft_model, am, [&](Token t) { out.push_back(t); });
        last = out.back();
    }
    out.resize(steps);
    return out;
}

} // namespace powerserve

// ---------------------------------------------------------------- C driver API
using namespace powerserve;
extern "C" {
extern const char *psh_last_error(void);
void psh_set_error(const char *msg);
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
// draft sampler alone (TopK -> Temperature -> Softmax): returns the number of entries written
int psh_draft_sample(const float *logits, int n, int top_k, float temperature, int32_t *tokens, float *probs) {
    try {
        const auto r = draft_sample(std::span<const float>(logits, (size_t)n), (size_t)top_k, temperature);
        for (size_t i = 0; i < r.size(); i++) { tokens[i] = r[i].token; probs[i] = r[i].prob; }
        return (int)r.size();
    } catch (const std::exception &e) { psh_set_error(e.what()); return -1; }
}
}
