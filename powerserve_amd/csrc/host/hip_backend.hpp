// HIPBackend: drop-in peer of ggml::GGMLBackend (src/backend/ggml/ggml.hpp:186-250) — same method names, arity
// and argument meaning — running on one MI355X through the C-ABI in include/ps_hip.h.  Platform mirrors
// src/backend/platform.hpp:28-51 with a `hip_backends` map next to where the reference keeps `ggml_backends`.
#pragma once
#include "../../../include/ps_hip.h"
#include "graph.hpp"
#include "json_gguf.hpp"

#include <map>

namespace powerserve {

struct Weight;
namespace hip {

// FP32 KV cache on the device in the reference layout (src/backend/ggml/ggml_kv_cache.cpp:35-57):
// K [n_ctx][kv_dim], V [kv_dim][n_ctx]; tensors are exposed with shape {n_ctx, kv_dim} exactly like
// GGMLKV::chunk.key_tensors so NormAttention::build's views apply unchanged.  Position bookkeeping follows
// KVCache<T>::advance_tokens / rollback / truncate (src/core/kv_cache.hpp:249-272).
struct HIPKV {
    size_t m_kv_dim, m_n_kv_heads, m_n_ctx, m_n_layers, m_head_size, m_batch_size = 1, kv_size = 0;
    std::vector<Tensor> key_tensors, value_tensors;
    ps_hip_model *m_model;
    HIPKV(const ModelConfig::LLMConfig &cfg, ps_hip_model *model);
    size_t position() const { return ps_hip_model_kv_position(m_model); }
    void reset_batch_size(size_t bs) { m_batch_size = bs; }
    void reset_kv_cache() { ps_hip_model_kv_truncate(m_model, kv_size); }
    void advance(int n);
    void rollback(size_t n);
    // the rest of KVCacheInterface (src/core/kv_cache.hpp:120-162) on the device cache.  A forward writes its K / V rows straight into the
    // slots position() + i (there is no per-batch staging area to copy out of), hence: copy = move from slot position() + token index,
    // save_tokens = bounds check only, unmask_tokens = un-hide the slots behind the position, append_tokens = the three in the reference's order.
    void copy(size_t dst_cache_index, size_t src_token_index);
    void move(size_t dst_cache_index, size_t src_cache_index);
    void mask(size_t cache_index);
    void unmask(size_t cache_index);
    void save_tokens(size_t n_tokens);
    void save_kv(int size) { save_tokens((size_t)size); } // GGMLKV::save_kv (ggml_kv_cache.hpp:148-150)
    void unmask_tokens(size_t n_tokens);
    size_t advance_tokens(size_t n_tokens) { const size_t old = position(); advance((int)n_tokens); return old; }
    size_t rollback_tokens(size_t n_tokens) { const size_t old = position(); rollback(n_tokens); return old; }
    size_t truncate_tokens(size_t n_tokens) { const size_t old = position(); ps_hip_model_kv_truncate(m_model, n_tokens); return old; }
    size_t append_tokens(size_t n_tokens);
    auto get_cache(size_t L) -> std::pair<Tensor &, Tensor &> { return {key_tensors[L], value_tensors[L]}; }
};

// What HIPBackend::plan compares a graph's weight operands with: the device handles the attached model was created from
// (ps_weight* for matrices, device float* for norm weights / biases), in the order NormAttention::build / FFN::build use them.
struct LoweringTable {
    struct Layer { const void *attn_norm, *wq, *wk, *wv, *bq, *bk, *bv, *wo, *ffn_norm, *wg, *wu, *wd; };
    const void *token_embd = nullptr, *output = nullptr, *output_norm = nullptr;
    std::vector<Layer> layers;
    bool bias = false;
};

struct HIPBackend {
    ps_hip_ctx *m_ctx = nullptr;
    ps_hip_model *m_model = nullptr; // fused fast path + owner of the KV cache
    std::unique_ptr<HIPKV> m_kv;
    ModelConfig::LLMConfig m_config;
    int m_device;
    bool m_fused = true; // plan(): lower the canonical layer sequence to the fused kernels

    HIPBackend(const ModelConfig::LLMConfig &config, const HyperParams &hparams, int device);
    ~HIPBackend();
    void attach_model(ps_hip_model *m, LoweringTable table = {}); // called once the weights are uploaded

    // ---- the reference's op set
    void add(const Tensor *dst, const Tensor *src0, const Tensor *src1) const;
    void get_embedding(const Tensor *dst, const Tensor *weight, const std::vector<int> &tokens) const;
    void matmul(const Tensor *dst, const Tensor *src0, const Tensor *src1) const;
    void rmsnorm(const Tensor *o, const Tensor *x, const Tensor *weight, float eps) const;
    void rope(Tensor *out, const Tensor *src, const std::vector<int> &pos, const ModelConfig::LLMConfig::RopeConfig &rope_cfg) const;
    void softmax(const Tensor *out, const Tensor *x) const;
    void permute(const Tensor *out, const Tensor *x, Shape axes) const;
    void cont(const Tensor *out, const Tensor *x) const;
    void softmax_ext(const Tensor *out, const Tensor *x, const Tensor *mask, float scale, float max_bias) const;
    bool is_contiguous(const Tensor *tensor, int n) const;
    // ggml_wrapper.cpp:225-270: how many threads the CPU pool gives an op -- what GGMLBackend::plan sizes its work buffer with.  A launch covers
    // an op whole, so every op is ONE task here; kept so that code written against the reference's backend compiles and sizes nothing.
    int get_n_tasks(std::shared_ptr<OpNode> op);
    int get_vec_dot_type(const Tensor *tensor) const;
    void silu_hadamard(const Tensor *out, const Tensor *hb, const Tensor *hb2) const;
    void copy(const Tensor *dst, const Tensor *src) const;
    void print(const Tensor *x, size_t size) const;
    void reset_kv_batch_size(size_t batch_size) const { m_kv->reset_batch_size(batch_size); }
    // ggml.cpp:153-168, deprecated there ("This function is deprecated!", no caller: the graphs append through VIEW + COPY): k, v are the
    // batch's rows (kv_dim, batch_size) on the device; they are copied behind the cache position of layer L like the reference's memcpy pair
    // (V into the transposed cache this backend's attention reads).  The position is not moved.
    void add_cache(const Tensor *k, const Tensor *v, size_t L, const std::vector<int> &pos, size_t head_id);
    void transpose(const Tensor *out, const Tensor *x) const;
    void get_mask(const Tensor *out, const std::vector<int> &pos, const CausalAttentionMask &mask) const; // executor.cpp:210-224
    // The reference hands the whole op vector to the backend before running it (executor.cpp:47-49,79).  Here that is the
    // fusion hook: a graph that is exactly the canonical forward of the attached model (get_embedding, L x [NormAttention,
    // FFN] as their build() emit them, optional final norm + lm_head) is lowered to the fused launch plan
    // (ps_hip_model_forward_lowered: 5 launches per layer for one token); anything else runs op by op.
    void plan(std::vector<std::shared_ptr<OpNode>> &ops);
    bool lowered() const { return m_low.ok; }
    void run_lowered();                 // Executor::run, lowered graph
    int n_plans = 0, n_lowered = 0;     // statistics (tests, bench): graphs handed to plan() / lowered by it, plan-only queries included ...
    int n_probes = 0;                   // ... which are also counted here (Model::prefill asking whether a chunk shape lowers)
    void discard_plan() {               // a plan-only query: nothing of it is retained -- lowered() and run_lowered() are about graphs that run
        n_probes++;
        m_low = Lowered{};
    }
    void setup_work_data(size_t) {}
    void setup_threadpool() {}  // a generation is bracketed by these in the reference (model.hpp:145,165-168):
    void reset_threadpool();    // here: nothing to create, drain the stream at the end
    void sync() const;

    // per-forward arena for graph intermediates (replaces malloc-per-tensor, src/executor/executor.cpp:23-45)
    void *arena_alloc(size_t bytes);
    void arena_reset() { m_arena_off = 0; }
    void arena_reserve(size_t bytes); // grows the arena (between forwards only)

private:
    struct Lowered { bool ok = false; std::vector<int32_t> tokens, pos; std::vector<uint8_t> tree; bool lm_head = false; Tensor *logits = nullptr; } m_low;
    LoweringTable m_table;
    bool match_canonical(std::vector<std::shared_ptr<OpNode>> &ops, Lowered &out) const;
    void check(int rc, const char *what) const;
    ps_tensor to_ps(const Tensor *t) const;
    char *m_arena = nullptr;
    size_t m_arena_cap = 0, m_arena_off = 0;
};

} // namespace hip

struct Platform {
    std::map<std::string, std::unique_ptr<hip::HIPBackend>> hip_backends;
    void init_hip_backend(const std::shared_ptr<ModelConfig> &config, const HyperParams &hparams, int device = 0);
    void destroy_hip_backend(const std::shared_ptr<ModelConfig> &config) { hip_backends.erase(config->model_id); }
    size_t get_kv_position(std::string &model_id) const { return hip_backends.at(model_id)->m_kv->position(); }
    void reset_kv_position(std::string &model_id) { hip_backends[model_id]->m_kv->reset_kv_cache(); }
};

struct Executor { // src/executor/executor.hpp:22-45
    Platform &m_platform;
    Graph &m_graph;
    Executor(Platform &platform, Graph &graph) : m_platform(platform), m_graph(graph) {}
    void allocate_buffers();
    void plan();
    void run();
    bool lowered() const { return m_platform.hip_backends.at(m_graph.m_model_id)->lowered(); } // after plan(): no intermediate buffers needed
private:
    bool m_planned = false;
};

} // namespace powerserve
