// Sampler chain, host side — mirrors
//   ProbIndex / ProbArray          src/sampler/prob_array.hpp:24-82, prob_array.cpp:21-67
//   Temperature / Softmax / Normalize / TopK / TopP / RepeatPenalty / Stochastic samplers
//                                  src/sampler/sampler.hpp:26-127, sampler.cpp:19-186
//   SamplerChain::build_from_config src/sampler/sampler_chain.cpp:19-51 (order: repeat penalty, top-k, temperature,
//                                  softmax, top-p, normalize, stochastic)
//   HyperParams::SamplerConfig     src/core/config.hpp:34-47
// Logits come from the backend (ps_hip_model_logits); this is plain host arithmetic on at most vocab_size floats per
// token, kept bit-compatible with the reference (same float / double mix, same std::mt19937 + discrete_distribution).
#pragma once
#include "core.hpp"

#include <deque>
#include <memory>
#include <random>
#include <span>
#include <vector>

namespace powerserve {

struct ProbIndex {
    float prob  = 0.0f;
    Token token = -1;
    bool operator<(const ProbIndex &o) const { return prob < o.prob; }
    bool operator>(const ProbIndex &o) const { return prob > o.prob; }
};

struct ProbArray {
    std::vector<ProbIndex> m_probs;
    bool m_is_sorted     = false; // descending
    bool m_is_normalized = false; // sums to 1
    explicit ProbArray(std::span<const float> logits);
    ProbIndex &operator[](size_t i) { return m_probs[i]; }
    void normalize();
    void softmax();
    void resize(size_t n) { m_probs.resize(n); }
    template <typename RandomEngine> ProbIndex &stochastic_sample(RandomEngine &&gen) {
        POWERSERVE_ASSERT(m_is_normalized);
        // weight i is evaluated at x = i + 0.5 (discrete_distribution(count, xmin, xmax, unary_op))
        const size_t index = std::discrete_distribution<size_t>(m_probs.size(), 0, m_probs.size(), [&](double x) { return m_probs[(size_t)x].prob; })(gen);
        return m_probs[index];
    }
    ProbIndex &greedy_sample();
};

struct SamplerConfig {
    uint64_t seed     = (uint64_t)-1; // -1: random_device
    float temperature = 0.80f;
    float top_p       = 0.95f;
    size_t top_k      = 40;
    size_t min_keep   = 0;
    int penalty_last_n    = 64;
    float penalty_repeat  = 1.00f;
    float penalty_freq    = 0.00f;
    float penalty_present = 0.00f;
    bool penalize_nl      = false;
    bool ignore_eos       = false;
};

struct Sampler {
    virtual ~Sampler() = default;
    virtual void apply(ProbArray &probs) = 0;
    virtual void accept(Token) {}
};
struct TemperatureSampler final : Sampler {
    float m_temperature;
    explicit TemperatureSampler(float t) : m_temperature(t) {}
    void apply(ProbArray &probs) override;
};
struct SoftmaxSampler final : Sampler { void apply(ProbArray &probs) override { probs.softmax(); } };
struct NormalizeSampler final : Sampler { void apply(ProbArray &probs) override { probs.normalize(); } };
struct TopKSampler final : Sampler {
    size_t m_topk;
    explicit TopKSampler(size_t k) : m_topk(k) {}
    void apply(ProbArray &probs) override;
};
struct TopPSampler final : Sampler {
    float m_topp;
    size_t m_min_keep;
    explicit TopPSampler(float p, size_t min_keep = 1) : m_topp(p), m_min_keep(min_keep) {}
    void apply(ProbArray &probs) override;
};
struct RepeatPenaltySampler final : Sampler {
    static constexpr Token null_token = -1;
    int32_t m_vocab_size;
    Token m_special_eos_id, m_linefeed_id;
    int32_t m_penalty_last_n;
    float m_penalty_repeat, m_penalty_freq, m_penalty_present;
    bool m_penalize_nl, m_ignore_eos;
    std::deque<Token> m_prev;
    RepeatPenaltySampler(int32_t vocab_size, Token special_eos_id, Token linefeed_id, int32_t penalty_last_n, float penalty_repeat, float penalty_freq,
                         float penalty_present, bool penalize_nl, bool ignore_eos);
    void apply(ProbArray &probs) override;
    void accept(Token token) override;
};
struct StochasticSampler final : Sampler {
    std::mt19937 m_random_state;
    explicit StochasticSampler(uint64_t seed) : m_random_state(seed) {}
    void apply(ProbArray &probs) override;
};

struct SamplerChain final : Sampler {
    SamplerChain() = default;
    // the two vocabulary facts the reference takes from its Tokenizer (sampler_chain.cpp:34-36)
    SamplerChain(const SamplerConfig &config, int32_t n_vocabs, Token special_eos_id, Token linefeed_id) { build_from_config(config, n_vocabs, special_eos_id, linefeed_id); }
    template <typename S, typename... Args> void append(Args &&...args) { m_samplers.emplace_back(std::make_unique<S>(std::forward<Args>(args)...)); }
    void build_from_config(const SamplerConfig &config, int32_t n_vocabs, Token special_eos_id, Token linefeed_id);
    void apply(ProbArray &probs) override;
    void accept(Token token) override;
    // apply + take probs[0] + accept: one sampling step as ModelTokenIterator::decode does (model.hpp:170-183)
    Token sample(std::span<const float> logits);
    uint64_t m_seed = 0;

private:
    std::vector<std::unique_ptr<Sampler>> m_samplers;
};

} // namespace powerserve
