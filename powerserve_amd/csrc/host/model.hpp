// Model layer: weights, graph builders and the token loop, mirroring
//   Weight / LayerWeights        src/model/common/weights.hpp:24-74, llama_weight.hpp:24-34, qwen2_weight.hpp:24-37
//   NormAttention::build          src/model/module/norm_attention.cpp:26-160
//   FFN::build                    src/model/module/ffn.cpp:22-42
//   LlamaModel / Qwen2Model       src/model/llama/llama_model.cpp:52-132, src/model/qwen2/qwen2_model.cpp:75
//   ModelTokenIterator            src/model/model.hpp:117-184
//   load_model                    src/model/model_loader.cpp:23-41
#pragma once
#include <set>
#include "hip_backend.hpp"
#include "sampler.hpp"

#include <deque>
#include <span>

namespace powerserve {

struct LayerWeights {
    Tensor attn_norm, ffn_norm, attn_q, attn_k, attn_v, attn_output, ffn_gate, ffn_up, ffn_down, attn_q_bias, attn_k_bias, attn_v_bias;
};
struct Weight {
    Tensor token_embedding_table, output_weight, rms_final_weight;
    std::vector<LayerWeights> lw;
    bool tied = false;
};

struct LogitsVector { // src/model/model.hpp:27-39
    BufferPtr buffer;
    std::vector<std::span<const float>> logits_vector;
    LogitsVector() = default;
    LogitsVector(BufferPtr buf, size_t vocab_size, size_t batch_size) : buffer(buf) {
        const float *l = static_cast<const float *>(dynamic_cast<CPUBuffer &>(*buffer).m_data);
        for (size_t i = 0; i < batch_size; i++, l += vocab_size) logits_vector.emplace_back(l, l + vocab_size);
    }
};

struct NormAttention {
    const ModelConfig::LLMConfig &m_config;
    std::shared_ptr<Weight> m_weights;
    NormAttention(const ModelConfig::LLMConfig &c, std::shared_ptr<Weight> w) : m_config(c), m_weights(std::move(w)) {}
    TensorNode *build(Graph &g, TensorNode *x, int64_t L, const TensorNode *k_cache, const TensorNode *v_cache, const std::vector<int> &pos,
                      const CausalAttentionMask &mask, bool is_need_bias = false);
};
struct FFN {
    const ModelConfig::LLMConfig &m_config;
    std::shared_ptr<Weight> m_weights;
    FFN(const ModelConfig::LLMConfig &c, std::shared_ptr<Weight> w) : m_config(c), m_weights(std::move(w)) {}
    TensorNode *build(Graph &g, TensorNode *attn_o, int64_t L);
};

struct Model {
    std::string m_filename;
    std::shared_ptr<ModelConfig> m_config;
    std::shared_ptr<Weight> m_weights;
    std::shared_ptr<NormAttention> m_attn;
    std::shared_ptr<FFN> m_ffn;
    std::shared_ptr<Platform> m_platform;
    bool m_is_need_bias = false; // Qwen2
    bool m_use_plan_cache = true;  // a (batch size, lm_head) shape whose graph has been lowered once runs its launches without a second graph (false: every forward builds + plans)
    std::set<std::pair<size_t, bool>> m_lowered_shapes;
    int n_plan_cache_hits = 0;
    bool m_use_fused    = true;  // HIPBackend::plan lowers the canonical graph to the fused launches; false: op-by-op (A/B, tests)

    Model(const std::string &model_dir, const std::shared_ptr<ModelConfig> &config, const std::shared_ptr<Platform> &platform, int device,
          size_t max_batch);
    virtual ~Model();

    // one forward over `tokens` at consecutive positions `pos`; lm_head: logits for every token
    auto forward(const std::vector<int> &tokens, const std::vector<int> &pos, const CausalAttentionMask &mask, bool lm_head = true) -> LogitsVector;
    // greedy (top_k = 1 == arg-max, src/sampler/prob_array.cpp:65-67).  A lowered graph returns the device arg-max (4 bytes per token come to the host,
    // the logits stay on the GPU: SURVEY a21); an op-by-op graph copies the logits and takes the first maximum on the host like the reference.
    auto decode(const std::vector<Token> &tokens, const std::vector<int> &pos, bool lm_head) -> std::vector<Token>;
    // ModelTokenIterator's prefill loop (src/model/model.hpp:147-163): forward(chunk of batch_size tokens, lm_head = false) + advance, chunk after chunk, for
    // tokens appended at the cache position.  The first chunk's graph goes through Executor::plan like any forward; when the backend lowers it, the LOOP is
    // lowered too (ps_hip_model_prefill: the same bits, several reference chunks per launch sequence), else every chunk runs through forward().
    void prefill(const std::vector<Token> &tokens, size_t batch_size);
    // ModelTokenIterator: prefill all but the last prompt token in chunks of batch_size (no lm_head), then `steps`
    // single-token greedy steps
    auto generate(const std::vector<Token> &prompt, int steps, size_t batch_size) -> std::vector<Token>;
    // the same loop with a sampler chain instead of arg-max: logits come back to the host every step (model.hpp:170-183)
    auto generate(const std::vector<Token> &prompt, int steps, size_t batch_size, SamplerChain &sampler) -> std::vector<Token>;

    hip::HIPBackend &backend() { return *m_platform->hip_backends[m_config->model_id]; }

private:
    std::unique_ptr<GGUFFile> m_gguf;
    std::vector<ps_weight *> m_dev_weights;
    std::vector<void *> m_dev_f32;
    // ids != nullptr (greedy callers): a lowered graph fills it from the device arg-max and returns no logits
    auto forward_graph(const std::vector<int> &tokens, const std::vector<int> &pos, const CausalAttentionMask &mask, bool lm_head,
                       std::vector<Token> *ids = nullptr, bool plan_only = false) -> LogitsVector;
    bool m_last_lowered = false; // the most recent forward_graph was lowered by HIPBackend::plan
};
using LlamaModel = Model;
using Qwen2Model = Model;

auto load_model(const std::string &model_dir, const std::shared_ptr<Platform> &platform, int device = 0, size_t max_batch = 128, int n_ctx_cap = 0)
    -> std::shared_ptr<Model>;

} // namespace powerserve

// handle behind the C driver API (psh_model_*, psh_spec_*)
struct psh_model {
    std::shared_ptr<powerserve::Platform> platform;
    std::shared_ptr<powerserve::Model> model;
    std::string err;
};
