// oracle/ref_token_tree.cpp — TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libps_ref.so).
//
// Drives the REAL reference token tree — src/speculative/token_tree.cpp, compiled in place by oracle/Makefile — with two
// scripted models, so that its draft / verify behaviour can be recorded as fixtures (oracle/gen_golden_spec.py ->
// tests/golden/token_tree.npz) and the product's TokenTree (powerserve_amd/csrc/host/speculative.cpp) checked against it
// node for node and KV call for KV call.
//
// What is mine here: ScriptedKV (a KVCacheInterface, src/core/kv_cache.hpp:97-162, that remembers WHICH token sits in
// which cache slot and whether the slot is visible, and logs every call), ScriptedModel (a Model, src/model/model.hpp:
// 80-113, whose logits are a hash of the set of (token, position) entries the new token can see — so a wrong mask, move
// or copy changes what the model "says" next), and the loop of SpecTokenIterator::generate_tokens
// (src/speculative/spec_model.hpp:92-111: draft, tree forward on the target, rollback, verify).  The Tokenizer the tree
// wants for should_stop() is the reference's own, loaded from a GGUF with tokenizer.ggml.model = "no_vocab" that the
// reference's gguf writer produces on the fly (llama-vocab.cpp:2223-2240: no special tokens, so nothing stops).
//
// The scripted logits are defined by integer arithmetic plus three float operations so that the Python twin in
// tests/spec_script.py reproduces them bit for bit:
//     mix(z):  z ^= z>>30; z *= 0xBF58476D1CE4E5B9; z ^= z>>27; z *= 0x94D049BB133111EB; z ^= z>>31
//     e(tok,pos) = mix(tok*0x9E3779B97F4A7C15 + pos*0xD1B54A32D192ED03 + 1)
//     ctx        = sum of e over the visible entries, the new token included            (mod 2^64)
//     unit(seed, ctx, v) = float(mix(seed ^ ctx ^ (v+1)*0x9E3779B97F4A7C15) >> 40) / 2^24
//     logit[v]   = shared_w * unit(shared_seed, ctx, v) + own_w * unit(own_seed, ctx, v)    (float, unfused)

#include "ggml.h"
#include "model/model.hpp"
#include "speculative/token_tree.hpp"

#include <cstdint>
#include <cstdio>
#include <unistd.h>

using namespace powerserve;

namespace {

enum Op : int32_t { FORWARD1 = 1, FORWARD_TREE = 2, COPY = 3, MOVE = 4, MASK = 5, UNMASK = 6, ADVANCE = 7, ROLLBACK = 8, FORWARD1_NO_LOGITS = 9 };

struct EventLog {
    int32_t *buf;
    int cap, n = 0;
    void add(int model, int op, int64_t a, int64_t b) {
        if (n < cap) { buf[4 * n] = model; buf[4 * n + 1] = op; buf[4 * n + 2] = (int32_t)a; buf[4 * n + 3] = (int32_t)b; }
        n++;
    }
};

inline uint64_t mix(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
inline uint64_t entry_hash(int tok, int pos) { return mix((uint64_t)(int64_t)tok * 0x9E3779B97F4A7C15ull + (uint64_t)(int64_t)pos * 0xD1B54A32D192ED03ull + 1); }
inline float unit(uint64_t seed, uint64_t ctx, int v) { return (float)(mix(seed ^ ctx ^ ((uint64_t)(v + 1) * 0x9E3779B97F4A7C15ull)) >> 40) * (1.0f / 16777216.0f); }

struct ScriptedKV final : KVCacheInterface {
    int id;
    EventLog &log;
    std::vector<int> tok, pos, batch_tok, batch_pos;
    std::vector<uint8_t> vis;
    ScriptedKV(int id, size_t n_ctx, EventLog &log) : KVCacheInterface(1, 1, n_ctx), id(id), log(log), tok(n_ctx, -1), pos(n_ctx, -1), vis(n_ctx, 0) {}

    auto key_data(KVPosition) const -> KVView override { POWERSERVE_ABORT("scripted KV has no data"); }
    auto value_data(KVPosition) const -> KVView override { POWERSERVE_ABORT("scripted KV has no data"); }
    auto key_entry(KVPosition) const -> KVView override { POWERSERVE_ABORT("scripted KV has no data"); }
    auto value_entry(KVPosition) const -> KVView override { POWERSERVE_ABORT("scripted KV has no data"); }

    void copy_for_layers(size_t, size_t, size_t dst, size_t src_token) override {
        log.add(id, COPY, dst, src_token);
        tok[dst] = batch_tok.at(src_token);
        pos[dst] = batch_pos.at(src_token);
    }
    void move(size_t dst, size_t src) override {
        log.add(id, MOVE, dst, src);
        tok[dst] = tok[src];
        pos[dst] = pos[src];
    }
    void mask(size_t i) override { log.add(id, MASK, i, 0); vis[i] = 0; }
    void unmask(size_t i) override { log.add(id, UNMASK, i, 0); vis[i] = 1; }
    void save_tokens_for_layers(size_t, size_t, size_t n) override {
        for (size_t i = 0; i < n; i++) { tok[position + i] = batch_tok.at(i); pos[position + i] = batch_pos.at(i); }
    }
    void unmask_tokens(size_t n) override {
        for (size_t i = 0; i < n; i++) vis[position + i] = 1;
    }
    size_t advance_tokens(size_t n) override {
        log.add(id, ADVANCE, n, 0);
        unmask_tokens(n);
        const size_t old = position;
        position += n;
        return old;
    }
    size_t rollback_tokens(size_t n) override {
        log.add(id, ROLLBACK, n, 0);
        POWERSERVE_ASSERT(n <= position);
        const size_t old = position;
        position -= n;
        for (size_t i = 0; i < n; i++) vis[position + i] = 0;
        return old;
    }
    size_t truncate_tokens(size_t n) override {
        const size_t old = position;
        if (n < position) rollback_tokens(position - n);
        return old;
    }
};

struct ScriptedModel final : Model {
    ScriptedKV kv;
    uint64_t shared_seed, own_seed;
    float shared_w, own_w;
    int vocab;
    std::vector<float> logits;
    ScriptedModel(int id, size_t n_ctx, EventLog &log, uint64_t shared_seed, uint64_t own_seed, float shared_w, float own_w, int vocab)
        : Model("scripted"), kv(id, n_ctx, log), shared_seed(shared_seed), own_seed(own_seed), shared_w(shared_w), own_w(own_w), vocab(vocab) {
        kv_cache = &kv;
    }

    auto forward(const std::vector<int> &tokens, const std::vector<int> &pos, const CausalAttentionMask &mask, bool lm_head) -> LogitsVector override {
        const size_t n = tokens.size(), base = kv.position;
        if (n == 1) kv.log.add(kv.id, lm_head ? FORWARD1 : FORWARD1_NO_LOGITS, tokens[0], pos[0]);
        else kv.log.add(kv.id, FORWARD_TREE, n, base);
        kv.batch_tok = tokens;
        kv.batch_pos = pos;
        uint64_t past = 0; // what every token of the batch sees of the cache
        for (size_t s = 0; s < base; s++)
            if (kv.vis[s]) past += entry_hash(kv.tok[s], kv.pos[s]);
        LogitsVector ret;
        if (lm_head) {
            logits.assign(n * (size_t)vocab, 0.f);
            for (size_t i = 0; i < n; i++) {
                uint64_t ctx = past;
                for (size_t j = 0; j < n; j++)
                    if (mask.not_masked(i, j)) ctx += entry_hash(tokens[j], pos[j]);
                float *row = logits.data() + i * (size_t)vocab;
                for (int v = 0; v < vocab; v++) {
                    const float a = shared_w * unit(shared_seed, ctx, v), b = own_w * unit(own_seed, ctx, v);
                    row[v] = a + b;
                }
                ret.logits_vector.push_back(std::span<const float>(row, (size_t)vocab));
            }
        }
        kv.save_tokens(n); // (the backend's KV bookkeeping of a forward: src/backend/ggml/ggml_kv_cache.hpp:149-155)
        kv.unmask_tokens(n);
        kv.position += n;
        return ret;
    }
    auto decode(Sampler &, const std::vector<Token>, const std::vector<int>, bool) -> std::vector<Token> override { POWERSERVE_ABORT("not scripted"); }
    auto generate(const Tokenizer &, Sampler &, const std::string &, int, size_t) -> std::shared_ptr<TokenIterator> override { POWERSERVE_ABORT("not scripted"); }
};

std::string write_no_vocab_gguf(int vocab) { // the reference's own GGUF writer (libs/ggml/src/ggml.c gguf_write_to_file)
    char path[] = "/tmp/ps_ref_novocab_XXXXXX";
    const int fd = mkstemp(path);
    POWERSERVE_ASSERT(fd >= 0);
    close(fd);
    gguf_context *g = gguf_init_empty();
    gguf_set_val_str(g, "general.architecture", "llama");
    gguf_set_val_str(g, "tokenizer.ggml.model", "no_vocab");
    gguf_set_val_str(g, "tokenizer.chat_template", "chatml");
    gguf_set_val_u32(g, "llama.vocab_size", (uint32_t)vocab);
    gguf_write_to_file(g, path, false);
    gguf_free(g);
    return path;
}

} // namespace

extern "C" {

struct ref_spec_config { // same layout as psh_spec_config (powerserve_amd/csrc/host/speculative.cpp)
    int32_t draft_batch_size, top_k, max_fan_out, early_stop;
    float temperature, p_base, min_prob;
};
struct ref_script { // the two scripted models
    uint64_t shared_seed, target_seed, draft_seed;
    float shared_w, target_w, draft_w;
    int32_t vocab, n_ctx;
};

// prefix: tokens both caches hold at positions 0..n_prefix-1 before the first iteration.
// tree rows per iteration: draft_batch_size x {token, position, parent}; masks: draft_batch_size^2 bytes per iteration.
// events: rows of {model (0 target, 1 draft), op, a, b}; *n_events may exceed event_cap (then the log was truncated).
int ref_token_tree_run(const ref_spec_config *c, const ref_script *s, const int32_t *prefix, int n_prefix, int32_t root_token, int n_iterations,
                       int32_t *out_tokens, int32_t *n_out, int32_t *tree, uint8_t *masks, int32_t *events, int event_cap, int32_t *n_events,
                       uint64_t *stats_unused) {
    (void)stats_unused;
    SpeculativeConfig cfg;
    cfg.draft_batch_size          = (size_t)c->draft_batch_size;
    cfg.draft_sampler.top_k       = (size_t)c->top_k;
    cfg.draft_sampler.temperature = c->temperature;
    cfg.draft_sampler.p_base      = c->p_base;
    cfg.token_tree.max_fan_out    = (size_t)c->max_fan_out;
    cfg.token_tree.min_prob       = c->min_prob;
    cfg.token_tree.early_stop     = c->early_stop != 0;

    EventLog log{events, event_cap};
    auto target = std::make_shared<ScriptedModel>(0, (size_t)s->n_ctx, log, s->shared_seed, s->target_seed, s->shared_w, s->target_w, s->vocab);
    auto draft  = std::make_shared<ScriptedModel>(1, (size_t)s->n_ctx, log, s->shared_seed, s->draft_seed, s->shared_w, s->draft_w, s->vocab);
    for (ScriptedModel *m : {target.get(), draft.get()}) {
        for (int i = 0; i < n_prefix; i++) { m->kv.tok[i] = prefix[i]; m->kv.pos[i] = i; m->kv.vis[i] = 1; }
        m->kv.position = (size_t)n_prefix;
    }
    const std::string vocab_path = write_no_vocab_gguf(s->vocab);
    Tokenizer tokenizer(vocab_path);
    unlink(vocab_path.c_str());
    SamplerChain greedy; // empty chain: verify() then takes ProbArray::greedy_sample of the raw logits
    TokenTree tree_obj(cfg);
    const ModelPtr target_model = target, draft_model = draft;

    const size_t bs = cfg.draft_batch_size;
    std::vector<Token> out;
    Token last = root_token;
    for (int it = 0; it < n_iterations; it++) {
        // SpecTokenIterator::generate_tokens, spec_model.hpp:92-105
        tree_obj.draft(draft_model, tokenizer, bs, last);
        const auto tmask = tree_obj.attention_mask();
        CausalAttentionMask mask(bs, tmask);
        auto ret = target_model->forward(tree_obj.tokens(), tree_obj.positions(), mask);
        target_model->kv_cache->rollback_tokens(bs);
        tree_obj.verify(target_model, draft_model, greedy, ret.logits_vector, [&](Token t) { out.push_back(t); });
        last = out.back();

        const auto toks = tree_obj.tokens(), poss = tree_obj.positions();
        for (size_t u = 0; u < bs; u++) {
            int parent = -1; // nodes are created after their parents: the nearest visible earlier node is the parent
            for (size_t x = 0; x < u; x++)
                if (tmask[u][x]) parent = (int)x;
            int32_t *row = tree + ((size_t)it * bs + u) * 3;
            row[0] = toks[u]; row[1] = poss[u]; row[2] = parent;
            for (size_t x = 0; x < bs; x++) masks[((size_t)it * bs + u) * bs + x] = tmask[u][x];
        }
    }
    memcpy(out_tokens, out.data(), out.size() * 4);
    *n_out = (int32_t)out.size();
    *n_events = log.n;
    return 0;
}

} // extern "C"
