#include "sampler.hpp"

#include <algorithm>
#include <cmath>
#include <unordered_map>

namespace powerserve {

ProbArray::ProbArray(std::span<const float> logits) {
    m_probs.resize(logits.size());
    for (size_t i = 0; i < logits.size(); i++) m_probs[i] = {logits[i], (Token)i};
}

void ProbArray::normalize() {
    if (m_is_normalized) return;
    double sum = 0.;
    for (const auto &p : m_probs) sum += p.prob;
    for (auto &p : m_probs) p.prob /= sum; // (float /= double: divided in double, rounded once)
    m_is_normalized = true;
}

void ProbArray::softmax() {
    POWERSERVE_ASSERT(m_probs.size() > 0);
    if (!m_is_sorted) {
        std::sort(m_probs.begin(), m_probs.end(), std::greater());
        m_is_sorted = true;
    }
    const float max_prob = m_probs[0].prob;
    double exp_prob_sum  = 0; // smallest to largest
    for (auto it = m_probs.rbegin(); it != m_probs.rend(); ++it) {
        it->prob = std::exp(it->prob - max_prob);
        exp_prob_sum += it->prob;
    }
    for (auto &p : m_probs) p.prob /= exp_prob_sum;
    m_is_normalized = true;
}

ProbIndex &ProbArray::greedy_sample() { return *std::max_element(m_probs.begin(), m_probs.end()); }

void TemperatureSampler::apply(ProbArray &probs) {
    POWERSERVE_ASSERT(m_temperature > 0);
    if (m_temperature != 1) {
        for (auto &p : probs.m_probs) p.prob /= m_temperature;
        probs.m_is_normalized = false;
    }
}

void TopKSampler::apply(ProbArray &probs) {
    POWERSERVE_ASSERT(m_topk > 0);
    const size_t k = std::min(m_topk, probs.m_probs.size());
    if (!probs.m_is_sorted) {
        std::partial_sort(probs.m_probs.begin(), probs.m_probs.begin() + k, probs.m_probs.end(), std::greater<ProbIndex>{});
        probs.m_is_sorted = true;
    }
    if (k != probs.m_probs.size()) probs.m_is_normalized = false;
    probs.m_probs.resize(k);
}

void TopPSampler::apply(ProbArray &probs) {
    if (m_topp >= 1.0f) return;
    POWERSERVE_ASSERT(probs.m_is_normalized);
    POWERSERVE_ASSERT(probs.m_is_sorted);
    float cum_sum   = 0.0f;
    size_t last_idx = probs.m_probs.size();
    for (size_t i = 0; i < probs.m_probs.size(); ++i) {
        cum_sum += probs.m_probs[i].prob;
        if (cum_sum >= m_topp && i + 1 >= m_min_keep) { last_idx = i + 1; break; }
    }
    if (last_idx != probs.m_probs.size()) probs.m_is_normalized = false;
    probs.m_probs.resize(last_idx);
}

RepeatPenaltySampler::RepeatPenaltySampler(int32_t vocab_size, Token special_eos_id, Token linefeed_id, int32_t penalty_last_n, float penalty_repeat,
                                           float penalty_freq, float penalty_present, bool penalize_nl, bool ignore_eos)
    : m_vocab_size(vocab_size), m_special_eos_id(special_eos_id), m_linefeed_id(linefeed_id), m_penalty_last_n(penalty_last_n),
      m_penalty_repeat(penalty_repeat), m_penalty_freq(penalty_freq), m_penalty_present(penalty_present), m_penalize_nl(penalize_nl),
      m_ignore_eos(ignore_eos) {
    if (linefeed_id == null_token) m_penalize_nl = true;
    if (special_eos_id == null_token) m_ignore_eos = false;
    // as in the reference the history starts as penalty_last_n entries of token 0 and only ever grows at the back, and the
    // window that is counted is its FIRST penalty_last_n entries (sampler.hpp:109-110, sampler.cpp:142-144)
    m_prev.resize(m_penalty_last_n);
}

void RepeatPenaltySampler::apply(ProbArray &probs) {
    auto find = [&](Token t) -> int64_t { // candidates not yet sorted / truncated: the token sits at its own index
        if (t >= 0 && probs.m_probs.size() > (size_t)t && probs.m_probs[t].token == t) return t;
        for (size_t i = 0; i < probs.m_probs.size(); ++i)
            if (probs.m_probs[i].token == t) return (int64_t)i;
        return -1;
    };
    if (m_ignore_eos) {
        const int64_t i = find(m_special_eos_id);
        if (i >= 0) probs.m_probs[i].prob = -INFINITY;
    }
    if (m_penalty_last_n == 0 || (m_penalty_repeat == 1.0f && m_penalty_freq == 0.0f && m_penalty_present == 0.0f)) return;
    int64_t nl_idx = -1;
    float nl_logit = -INFINITY;
    if (!m_penalize_nl) {
        POWERSERVE_ASSERT(m_linefeed_id >= 0);
        nl_idx = find(m_linefeed_id);
        if (nl_idx >= 0) nl_logit = probs.m_probs[nl_idx].prob;
    }
    std::unordered_map<Token, int> token_count;
    for (int i = 0; i < std::min<int>(m_penalty_last_n, (int)m_prev.size()); ++i) token_count[m_prev[i]]++;
    for (auto &p : probs.m_probs) {
        const auto it = token_count.find(p.token);
        if (it == token_count.end()) continue;
        const int count = it->second;
        if (p.prob <= 0) p.prob *= m_penalty_repeat; // multiply negative logits, divide positive ones
        else p.prob /= m_penalty_repeat;
        p.prob -= float(count) * m_penalty_freq + float(count > 0) * m_penalty_present;
    }
    probs.m_is_sorted = false;
    if (!m_penalize_nl && nl_idx >= 0) probs.m_probs[nl_idx].prob = nl_logit;
}

void RepeatPenaltySampler::accept(Token token) {
    if (m_penalty_last_n > 0) m_prev.push_back(token);
}

void StochasticSampler::apply(ProbArray &probs) {
    probs[0] = probs.stochastic_sample(m_random_state);
    probs.resize(1);
    probs[0].prob         = 1.0f;
    probs.m_is_sorted     = true;
    probs.m_is_normalized = true;
}

void SamplerChain::build_from_config(const SamplerConfig &config, int32_t n_vocabs, Token special_eos_id, Token linefeed_id) {
    uint64_t seed = config.seed;
    if (seed == (uint64_t)-1) {
        std::random_device rd;
        seed = rd();
    }
    m_seed = seed;
    append<RepeatPenaltySampler>(n_vocabs, special_eos_id, linefeed_id, config.penalty_last_n, config.penalty_repeat, config.penalty_freq,
                                 config.penalty_present, config.penalize_nl, config.ignore_eos);
    append<TopKSampler>(config.top_k);
    append<TemperatureSampler>(config.temperature);
    append<SoftmaxSampler>();
    append<TopPSampler>(config.top_p);
    append<NormalizeSampler>();
    append<StochasticSampler>(seed);
}

void SamplerChain::apply(ProbArray &probs) {
    for (auto &s : m_samplers) s->apply(probs);
}
void SamplerChain::accept(Token token) {
    for (auto &s : m_samplers) s->accept(token);
}
Token SamplerChain::sample(std::span<const float> logits) {
    ProbArray probs(logits);
    apply(probs);
    const Token next = probs[0].token;
    accept(next);
    return next;
}

} // namespace powerserve
