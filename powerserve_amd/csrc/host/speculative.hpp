// Speculative decoding with a token tree, host side — mirrors
//   SpeculativeConfig             src/speculative/speculative_config.hpp:21-36
//   TokenTree (draft / verify)    src/speculative/token_tree.hpp:30-120, token_tree.cpp:35-39,60-94,96-234,279-315
//   SpeculativeModel / iterator   src/speculative/spec_model.hpp:21-114
// over the C-ABI of the HIP backend (ps_hip_model_forward_tree, _kv_mask, _kv_move, _kv_advance, _kv_rollback).
// The draft model grows a tree of candidate continuations with single-token forwards (branches are switched by hiding
// and showing its own cache slots); the target model scores the whole tree in ONE batched forward with a tree attention
// mask and per-node RoPE positions; the longest path the target agrees with is kept by moving its KV entries into place.
#pragma once
#include "model.hpp"
#include "sampler.hpp"

#include <functional>
#include <queue>

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

// TopK -> Temperature -> Softmax over one logits row (sampler.cpp:19-58, prob_array.cpp:37-59): sorted by probability
std::vector<ProbIndex> draft_sample(std::span<const float> logits, size_t top_k, float temperature);

struct TokenTree {
    static constexpr int NO_PARENT    = -1;
    static constexpr int NOT_IN_CACHE = -1;
    struct Node {
        int parent = NO_PARENT, depth = 0;
        Token token      = 0;
        int position     = 0;
        int cache_index  = NOT_IN_CACHE;
        float current_prob = 1.0f;
        bool accepted    = false;
        std::vector<int> children;
    };
    struct Stat {
        size_t n_draft_times = 0, n_draft_tokens = 0, n_accepted_tokens = 0, n_iterations = 0, n_generated_tokens = 0;
    };

    SpeculativeConfig m_config;
    std::vector<Node> m_nodes;
    Stat m_stat;

    explicit TokenTree(const SpeculativeConfig &config) : m_config(config) {}

    std::vector<int32_t> tokens() const;
    std::vector<int32_t> positions() const;
    std::vector<uint8_t> attention_mask() const; // [n][n]: node u sees its ancestors and itself

    void draft(Model &draft_model, size_t batch_size, Token root_token, const std::function<bool(Token)> &should_stop = {});
    // target_argmax[u]: the target's greedy token after node u (from the tree forward)
    void verify(Model &target_model, Model &draft_model, const std::vector<int32_t> &target_argmax, const std::function<void(Token)> &enqueue);

private:
    int lca(int u, int v) const;
    void switch_parent(Model &draft_model, int old_parent, int new_parent);
};

struct SpeculativeModel {
    std::shared_ptr<Model> target_model, draft_model;
    SpeculativeConfig config;
    TokenTree token_tree;
    SpeculativeModel(std::shared_ptr<Model> target, std::shared_ptr<Model> draft, const SpeculativeConfig &cfg = {})
        : target_model(std::move(target)), draft_model(std::move(draft)), config(cfg), token_tree(cfg) {}
    // prefill both models with all but the last prompt token, then draft / verify iterations until `steps` tokens exist
    std::vector<Token> generate(const std::vector<Token> &prompt, int steps, size_t batch_size);
};

} // namespace powerserve
