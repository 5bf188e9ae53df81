// Token sampling on the host.  What has to agree with the reference (pinned token for token by
// tests/test_sampler_vs_ref.py against the reference's own classes in oracle/_ref):
//   candidate list + its two invariants        src/sampler/prob_array.hpp:24-82, prob_array.cpp:21-67
//   the arithmetic of each stage               src/sampler/sampler.cpp:19-186
//   stage order of a configured chain          src/sampler/sampler_chain.cpp:19-51
//   configuration fields and defaults          src/core/config.hpp:34-47
// How it is organised here: a chain is a flat list of stage closures over one candidate list; the stages themselves are
// free functions in namespace `stage` (usable on their own — the token tree's draft sampler is three of them), the only
// stateful pieces (repeat-penalty history, the random engine) are owned by the chain.  `Sampler` stays as the plug-in
// interface (apply / accept) so a caller can still splice its own stage into a chain.
#pragma once
#include "core.hpp"

#include <functional>
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

// Candidates of one sampling step.  `m_is_sorted`: descending by prob; `m_is_normalized`: probs sum to one.
struct ProbArray {
    std::vector<ProbIndex> m_probs;
    bool m_is_sorted     = false;
    bool m_is_normalized = false;
    explicit ProbArray(std::span<const float> logits);
    ProbIndex &operator[](size_t i) { return m_probs[i]; }
    size_t size() const { return m_probs.size(); }
    void resize(size_t n) { m_probs.resize(n); }
    void sort_descending(size_t first_n); // only the leading first_n entries are put in order
    void normalize();
    void softmax();
    ProbIndex &greedy_sample();
    template <typename Engine> ProbIndex &stochastic_sample(Engine &&gen) {
        POWERSERVE_ASSERT(m_is_normalized);
        // discrete_distribution(count, xmin, xmax, op) evaluates op at the bucket centres i + 0.5
        std::discrete_distribution<size_t> pick(size(), 0, size(), [this](double x) { return m_probs[(size_t)x].prob; });
        return m_probs[pick(gen)];
    }
};

namespace stage {
void top_k(ProbArray &c, size_t k);
void temperature(ProbArray &c, float t);
inline void softmax(ProbArray &c) { c.softmax(); }
void top_p(ProbArray &c, float p, size_t min_keep = 1);
inline void normalize(ProbArray &c) { c.normalize(); }
void draw(ProbArray &c, std::mt19937 &engine); // collapses the list to the drawn candidate

// Logit penalties for recently seen tokens.  Reference behaviour kept as is: the history starts as `last_n` zeros and is
// appended to, and the window that gets counted is its first `last_n` entries.
struct RepeatPenalty {
    static constexpr Token none = -1;
    Token eos = none, linefeed = none;
    int last_n = 0;
    float repeat = 1.f, freq = 0.f, present = 0.f;
    bool spare_linefeed = false, ban_eos = false;
    std::vector<Token> history;
    RepeatPenalty() = default;
    RepeatPenalty(const SamplerConfig &cfg, Token eos_id, Token linefeed_id);
    void apply(ProbArray &c) const;
    void accept(Token t) { if (last_n > 0) history.push_back(t); }
    bool active() const { return last_n != 0 && !(repeat == 1.0f && freq == 0.0f && present == 0.0f); }
};
} // namespace stage

struct Sampler { // plug-in interface
    virtual ~Sampler() = default;
    virtual void apply(ProbArray &probs) = 0;
    virtual void accept(Token) {}
};

struct SamplerChain final : Sampler {
    using Stage = std::function<void(ProbArray &)>;
    SamplerChain() = default;
    SamplerChain(const SamplerChain &) = delete; // the configured stages refer to the chain's own state
    SamplerChain &operator=(const SamplerChain &) = delete;
    // n_vocabs / special_eos_id / linefeed_id are the vocabulary facts the reference reads from its Tokenizer
    SamplerChain(const SamplerConfig &config, int32_t n_vocabs, Token special_eos_id, Token linefeed_id) { build_from_config(config, n_vocabs, special_eos_id, linefeed_id); }
    void build_from_config(const SamplerConfig &config, int32_t n_vocabs, Token special_eos_id, Token linefeed_id);
    void append(Stage s) { m_stages.push_back(std::move(s)); }
    void append(std::shared_ptr<Sampler> s); // a user stage: apply in place, accept forwarded
    void apply(ProbArray &probs) override;
    void accept(Token token) override;
    Token sample(std::span<const float> logits); // apply, take candidate 0, accept it
    uint64_t m_seed = 0;

private:
    std::vector<Stage> m_stages;
    std::vector<std::shared_ptr<Sampler>> m_plugins;
    stage::RepeatPenalty m_penalty;
    std::mt19937 m_engine;
};

} // namespace powerserve
