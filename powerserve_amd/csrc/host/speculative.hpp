// Speculative decoding with a token tree, host side.  Behaviour follows
//   SpeculativeConfig             src/speculative/speculative_config.hpp:21-36
//   TokenTree (draft / verify)    src/speculative/token_tree.hpp:30-120, token_tree.cpp:35-39,96-282,358-377
//   SpeculativeModel / iterator   src/speculative/spec_model.hpp:21-114
// and is pinned to the reference's own token_tree.cpp (compiled into oracle/_ref, driven by scripted models) by
// tests/test_token_tree_vs_ref.py: same node order, parents, positions, KV call sequence and emitted tokens.
//
// The tree talks to its two models through SpecBackend — seven calls: a single-token forward, a masked tree forward,
// and five KV-slot operations.  HIPSpecBackend puts them on the C-ABI of the HIP backend (ps_hip_model_forward_tree,
// _kv_mask, _kv_move, _kv_advance, _kv_rollback); CallbackSpecBackend forwards them to plain C function pointers
// (psh_spec_backend below), which is how the CPU test drives the tree with scripted logits.
//
// The draft model grows a tree of candidate continuations with single-token forwards (branches are switched by hiding
// and showing its own cache slots); the target model scores the whole tree in ONE batched forward with a tree attention
// mask and per-node RoPE positions; the longest path the target agrees with is kept by moving its KV entries into place.
#pragma once
#include "model.hpp"
#include "sampler.hpp"

#include <functional>
#include <queue>

extern "C" {
// A model as the token tree sees it.  Every function returns 0 on success (kv_position: the position).
typedef struct psh_spec_backend {
    void *user;
    int64_t (*kv_position)(void *user);
    // one token at the current cache slot with RoPE position `position`; the slot becomes visible and the cache position
    // moves on by one.  logits: vocab_size floats to fill, or NULL when the caller does not need them
    int (*forward_one)(void *user, int32_t token, int32_t position, float *logits);
    // n tokens at cache slots position()..+n-1, node u attending to the visible past and to the nodes v with
    // mask[u*n+v] != 0; argmax[u] = the model's greedy token after node u.  The cache position does NOT move
    int (*forward_tree)(void *user, const int32_t *tokens, int32_t n, const int32_t *positions, const uint8_t *mask, int32_t *argmax);
    int (*kv_mask)(void *user, int64_t slot, int32_t visible);
    int (*kv_move)(void *user, int64_t dst_slot, int64_t src_slot);
    int (*kv_advance)(void *user, int64_t n);
    int (*kv_rollback)(void *user, int64_t n);
    int32_t vocab_size;
} psh_spec_backend;
}

namespace powerserve {

struct SpeculativeConfig {
    size_t draft_batch_size = 12;
    struct {
        size_t top_k      = 15;
        float temperature = 1.5f;
        float p_base      = 0.9f;
    } draft_sampler;
    struct {
        size_t max_fan_out = 3;
        float min_prob     = 0.2f;
        bool early_stop    = true;
    } token_tree;
};

struct SpecBackend {
    virtual ~SpecBackend() = default;
    virtual size_t kv_position() = 0;
    virtual void forward_one(Token token, int position, std::vector<float> *logits) = 0;
    virtual void forward_tree(const std::vector<int32_t> &tokens, const std::vector<int32_t> &positions, const std::vector<uint8_t> &mask,
                              std::vector<int32_t> &argmax) = 0;
    virtual void kv_mask(size_t slot, bool visible) = 0;
    virtual void kv_move(size_t dst_slot, size_t src_slot) = 0;
    virtual void kv_advance(size_t n) = 0;
    virtual void kv_rollback(size_t n) = 0;
    // the logits [n][vocab] of the most recent forward_tree, for a verify that samples (token_tree.cpp:214-216); false: this
    // backend only reports arg-max ids
    virtual bool tree_logits(size_t n, std::vector<float> &logits) { (void)n; (void)logits; return false; }
    virtual size_t vocab_size() const { return 0; }
    virtual size_t n_ctx() const { return 0; } // 0: unknown
};

struct HIPSpecBackend final : SpecBackend {
    Model &m;
    explicit HIPSpecBackend(Model &model) : m(model) {}
    size_t kv_position() override;
    void forward_one(Token token, int position, std::vector<float> *logits) override;
    void forward_tree(const std::vector<int32_t> &tokens, const std::vector<int32_t> &positions, const std::vector<uint8_t> &mask,
                      std::vector<int32_t> &argmax) override;
    void kv_mask(size_t slot, bool visible) override;
    void kv_move(size_t dst_slot, size_t src_slot) override;
    void kv_advance(size_t n) override;
    void kv_rollback(size_t n) override;
    bool tree_logits(size_t n, std::vector<float> &logits) override;
    size_t vocab_size() const override;
    size_t n_ctx() const override;

private:
    void check(int rc, const char *what);
};

struct CallbackSpecBackend final : SpecBackend {
    psh_spec_backend cb;
    explicit CallbackSpecBackend(const psh_spec_backend &c) : cb(c) {}
    size_t kv_position() override;
    void forward_one(Token token, int position, std::vector<float> *logits) override;
    void forward_tree(const std::vector<int32_t> &tokens, const std::vector<int32_t> &positions, const std::vector<uint8_t> &mask,
                      std::vector<int32_t> &argmax) override;
    void kv_mask(size_t slot, bool visible) override;
    void kv_move(size_t dst_slot, size_t src_slot) override;
    void kv_advance(size_t n) override;
    void kv_rollback(size_t n) override;
};

// the tree's draft sampler — top-k, temperature, softmax (token_tree.cpp:35-39): candidates sorted by probability
std::vector<ProbIndex> draft_sample(std::span<const float> logits, size_t top_k, float temperature);

struct TokenTree {
    static constexpr int NO_PARENT    = -1;
    static constexpr int NOT_IN_CACHE = -1;
    struct Node {
        int parent = NO_PARENT, depth = 0;
        Token token      = 0;
        int position     = 0;
        int cache_index  = NOT_IN_CACHE; // draft-model cache slot that holds this node's KV, if it was expanded
        float current_prob = 1.0f;
        bool accepted    = false;
        std::vector<int> children;
    };
    struct Stat {
        size_t n_draft_times = 0, n_draft_tokens = 0, n_accepted_tokens = 0, n_iterations = 0, n_generated_tokens = 0;
    };

    SpeculativeConfig m_config;
    std::vector<Node> m_nodes; // creation order = order of decreasing path probability at the time of creation
    Stat m_stat;

    explicit TokenTree(const SpeculativeConfig &config) : m_config(config) {}

    std::vector<int32_t> tokens() const;
    std::vector<int32_t> positions() const;
    std::vector<uint8_t> attention_mask() const; // [n][n]: node u sees its ancestors and itself

    void draft(SpecBackend &draft_model, size_t batch_size, Token root_token, const std::function<bool(Token)> &should_stop = {});
    // choose(u): the token the target model's sampler picks from node u's logits (greedy: its arg-max)
    void verify(SpecBackend &target_model, SpecBackend &draft_model, const std::function<Token(int)> &choose, const std::function<void(Token)> &enqueue);
    // target_argmax[u]: the target's greedy token after node u (from the tree forward)
    void verify(SpecBackend &target_model, SpecBackend &draft_model, const std::vector<int32_t> &target_argmax, const std::function<void(Token)> &enqueue) {
        verify(target_model, draft_model, [&](int u) { return (Token)target_argmax[(size_t)u]; }, enqueue);
    }
    // one draft / tree-forward / verify round starting from `last`; emitted tokens are appended to `out`.  sampler == nullptr:
    // greedy through the device arg-max (4 bytes per node cross the bus).  Otherwise every node on the accepted path goes
    // ProbArray -> sampler->apply -> greedy_sample exactly as token_tree.cpp:214-216 (the backend must provide tree_logits);
    // sampler->accept() is never called on this path -- the reference's speculative iterator does not call it either (only
    // Model::decode does, llama_model.cpp:128): penalties see the chain's initial history throughout
    void iterate(SpecBackend &target_model, SpecBackend &draft_model, Token last, std::vector<Token> &out, Sampler *sampler = nullptr,
                 const std::function<bool(Token)> &should_stop = {});

private:
    int lca(int u, int v) const;
    void switch_parent(SpecBackend &draft_model, int old_parent, int new_parent);
};

struct SpeculativeModel {
    std::shared_ptr<Model> target_model, draft_model;
    SpeculativeConfig config;
    TokenTree token_tree;
    SpeculativeModel(std::shared_ptr<Model> target, std::shared_ptr<Model> draft, const SpeculativeConfig &cfg = {})
        : target_model(std::move(target)), draft_model(std::move(draft)), config(cfg), token_tree(cfg) {}
    // prefill both models with all but the last prompt token, then draft / verify iterations until `steps` tokens exist
    // sampler: the target's sampler chain (hparams); null or pure greedy (top_k = 1, no penalties): the device arg-max.  Generation
    // ends early when the sampler's chain says so through should_stop (EOS without ignore_eos) or when the next tree would not
    // fit the cache window (position + draft_batch_size > n_ctx): the tokens so far are returned
    std::vector<Token> generate(const std::vector<Token> &prompt, int steps, size_t batch_size, Sampler *sampler = nullptr,
                                const std::function<bool(Token)> &should_stop = {});
};

} // namespace powerserve
