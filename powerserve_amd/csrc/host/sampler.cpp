#include "sampler.hpp"

#include <algorithm>
#include <cmath>

namespace powerserve {

ProbArray::ProbArray(std::span<const float> logits) : m_probs(logits.size()) {
    Token id = 0;
    for (auto &c : m_probs) { c.prob = logits[(size_t)id]; c.token = id++; }
}

void ProbArray::sort_descending(size_t first_n) {
    if (m_is_sorted) return;
    first_n = std::min(first_n, size());
    // (partial_sort over the full length is the library's heap sort; the reference sorts fully with std::sort only in
    //  softmax, see there)
    std::partial_sort(m_probs.begin(), m_probs.begin() + first_n, m_probs.end(), std::greater<ProbIndex>());
    m_is_sorted = true;
}

void ProbArray::normalize() {
    if (m_is_normalized) return;
    double total = 0.;
    for (const ProbIndex &c : m_probs) total += c.prob;
    for (ProbIndex &c : m_probs) c.prob = (float)((double)c.prob / total);
    m_is_normalized = true;
}

void ProbArray::softmax() {
    POWERSERVE_ASSERT(!m_probs.empty());
    if (!m_is_sorted) { // full order, std::sort as in prob_array.cpp:40 (tie order is the algorithm's)
        std::sort(m_probs.begin(), m_probs.end(), std::greater<ProbIndex>());
        m_is_sorted = true;
    }
    const float top = m_probs.front().prob;
    double total = 0.; // accumulated from the small end
    for (size_t i = size(); i-- > 0;) {
        const float e = std::exp(m_probs[i].prob - top);
        m_probs[i].prob = e;
        total += e;
    }
    for (ProbIndex &c : m_probs) c.prob = (float)((double)c.prob / total);
    m_is_normalized = true;
}

ProbIndex &ProbArray::greedy_sample() {
    size_t best = 0;
    for (size_t i = 1; i < size(); i++)
        if (m_probs[best].prob < m_probs[i].prob) best = i; // first of equals wins
    return m_probs[best];
}

namespace stage {

void top_k(ProbArray &c, size_t k) {
    POWERSERVE_ASSERT(k > 0);
    k = std::min(k, c.size());
    c.sort_descending(k);
    if (k < c.size()) {
        c.resize(k);
        c.m_is_normalized = false;
    }
}

void temperature(ProbArray &c, float t) {
    POWERSERVE_ASSERT(t > 0);
    if (t == 1) return;
    for (ProbIndex &x : c.m_probs) x.prob /= t;
    c.m_is_normalized = false;
}

void top_p(ProbArray &c, float p, size_t min_keep) {
    if (p >= 1.0f) return;
    POWERSERVE_ASSERT(c.m_is_normalized && c.m_is_sorted);
    float mass  = 0.0f;
    size_t keep = 0;
    while (keep < c.size()) {
        mass += c[keep++].prob;
        if (mass >= p && keep >= min_keep) break;
    }
    if (keep < c.size()) {
        c.resize(keep);
        c.m_is_normalized = false;
    }
}

void draw(ProbArray &c, std::mt19937 &engine) {
    const ProbIndex chosen = c.stochastic_sample(engine);
    c.resize(1);
    c[0] = {1.0f, chosen.token};
    c.m_is_sorted = c.m_is_normalized = true;
}

RepeatPenalty::RepeatPenalty(const SamplerConfig &cfg, Token eos_id, Token linefeed_id)
    : eos(eos_id), linefeed(linefeed_id), last_n(cfg.penalty_last_n), repeat(cfg.penalty_repeat), freq(cfg.penalty_freq),
      present(cfg.penalty_present), spare_linefeed(!cfg.penalize_nl && linefeed_id != none), ban_eos(cfg.ignore_eos && eos_id != none),
      history((size_t)std::max(cfg.penalty_last_n, 0), 0) {}

// position of `t` in the list: its own index while nothing has reordered the list yet, otherwise searched for
static ProbIndex *locate(ProbArray &c, Token t) {
    if (t >= 0 && (size_t)t < c.size() && c[(size_t)t].token == t) return &c[(size_t)t];
    for (ProbIndex &x : c.m_probs)
        if (x.token == t) return &x;
    return nullptr;
}

void RepeatPenalty::apply(ProbArray &c) const {
    if (ban_eos)
        if (ProbIndex *e = locate(c, eos)) e->prob = -INFINITY;
    if (!active()) return;
    ProbIndex *nl        = spare_linefeed ? locate(c, linefeed) : nullptr;
    const float nl_logit = nl ? nl->prob : 0.f;
    // occurrences inside the window, dense over the token ids that can appear in the list
    const size_t window = std::min((size_t)last_n, history.size());
    Token hi = -1;
    for (size_t i = 0; i < window; i++) hi = std::max(hi, history[i]);
    std::vector<int> seen((size_t)(hi + 1), 0);
    for (size_t i = 0; i < window; i++)
        if (history[i] >= 0) seen[(size_t)history[i]]++;
    for (ProbIndex &x : c.m_probs) {
        const int n = (x.token >= 0 && x.token <= hi) ? seen[(size_t)x.token] : 0;
        if (n == 0) continue;
        // a positive logit is divided, a non-positive one multiplied, so the token always becomes less likely
        x.prob = x.prob > 0 ? x.prob / repeat : x.prob * repeat;
        x.prob -= float(n) * freq + 1.0f * present;
    }
    c.m_is_sorted = false;
    if (nl) nl->prob = nl_logit;
}

} // namespace stage

void SamplerChain::append(std::shared_ptr<Sampler> s) {
    m_plugins.push_back(s);
    m_stages.push_back([s](ProbArray &c) { s->apply(c); });
}

void SamplerChain::build_from_config(const SamplerConfig &config, int32_t, Token special_eos_id, Token linefeed_id) {
    m_seed = config.seed;
    if (m_seed == (uint64_t)-1) m_seed = std::random_device{}();
    m_engine.seed((std::mt19937::result_type)m_seed);
    m_penalty = stage::RepeatPenalty(config, special_eos_id, linefeed_id);
    const float t = config.temperature, p = config.top_p;
    const size_t k = config.top_k;
    m_stages.clear();
    append([this](ProbArray &c) { m_penalty.apply(c); });
    append([k](ProbArray &c) { stage::top_k(c, k); });
    append([t](ProbArray &c) { stage::temperature(c, t); });
    append(stage::softmax);
    append([p](ProbArray &c) { stage::top_p(c, p); });
    append(stage::normalize);
    append([this](ProbArray &c) { stage::draw(c, m_engine); });
}

void SamplerChain::apply(ProbArray &probs) {
    for (const Stage &s : m_stages) s(probs);
}

void SamplerChain::accept(Token token) {
    m_penalty.accept(token);
    for (auto &s : m_plugins) s->accept(token);
}

Token SamplerChain::sample(std::span<const float> logits) {
    ProbArray candidates(logits);
    apply(candidates);
    const Token chosen = candidates[0].token;
    accept(chosen);
    return chosen;
}

} // namespace powerserve
