// Operator graph — the API a model builder programs against.  The builder methods, their argument meaning and their
// shape rules are the reference's (src/graph/graph.hpp:30-72 with the rules of graph.cpp:20-266; op set of
// op_type.hpp:19-45, parameter records of op_params.hpp:41-117), because src/model/module/*.cpp-style builders must run
// unchanged on this backend.  The representation underneath is this backend's own:
//   * an op is a flat record {kind, input tensors, output tensors, attributes}; attributes are a std::variant, not a
//     class hierarchy, and there is no bipartite node graph — program order in Graph::ops IS the schedule, and a tensor
//     only remembers the op that writes it;
//   * a view is a TensorNode whose `alias_of` names the tensor whose storage it shares;
//   * every builder method is one call of Graph::emit with the inferred result shape.
// HIPBackend::plan pattern-matches Graph::ops (hip_backend.cpp) and the Executor walks it op by op otherwise.
#pragma once
#include "core.hpp"

#include <variant>

namespace powerserve {

enum class OpType {
    NONE = 0, ADD, MAT_MUL, RMS_NORM, SILU_HADAMARD, ROPE, SOFTMAX, COPY, PRINT, GET_EMBEDDING, ADD_CACHE, PERMUTE, CONT, VIEW,
    SOFTMAX_EXT, GET_MASK, TRANSPOSE,
};

struct CausalAttentionMask { // src/model/module/attention_mask.hpp:43-50: causal unless an explicit (tree) mask is given
    size_t size = 0;
    std::vector<std::vector<bool>> mask;
    explicit CausalAttentionMask(size_t n) : size(n) {}
    CausalAttentionMask(size_t n, const std::vector<std::vector<bool>> &m) : size(n), mask(m) {}
    bool not_masked(size_t i, size_t j) const { return mask.empty() ? i >= j : mask[i][j]; }
};

// ---- per-op attributes
struct GetEmbeddingParams { std::vector<int> tokens; };
struct RMSNormParams { float eps; };
struct RopeParams { std::vector<int> pos; ModelConfig::LLMConfig::RopeConfig rope_cfg; };
struct AddCacheParams { size_t L; std::vector<int> pos; size_t head_id; };
struct PrintParams { size_t size = 0; };
struct PermuteParams { Shape axes; };
struct ViewParams { Shape stride; size_t offset; };
struct SoftmaxExtParams { float scale; float max_bias; };
struct GetMaskParams { const CausalAttentionMask *mask; std::vector<int> pos; }; // the mask object outlives the graph (one forward)
using OpAttr = std::variant<std::monostate, GetEmbeddingParams, RMSNormParams, RopeParams, AddCacheParams, PrintParams, PermuteParams, ViewParams,
                            SoftmaxExtParams, GetMaskParams>;

struct OpNode;

struct TensorNode : Tensor {
    OpNode *producer           = nullptr; // the op that writes this tensor; null for weights, caches and other graph inputs
    const TensorNode *alias_of = nullptr; // non-null: a view, no storage of its own
    TensorNode(const Tensor &t) : Tensor(t) {}
    TensorNode(DataType dt, const Shape &s) : Tensor(dt, s) {}
    bool is_view() const { return alias_of != nullptr; }
};
using TensorViewNode = TensorNode; // (the reference returns a distinct view-node type from view / permute / transpose)

struct OpNode {
    OpType op;
    std::vector<TensorNode *> in, out;
    OpAttr attr;
    explicit OpNode(OpType kind) : op(kind) {}
    template <typename T> const T &get_params() const {
        const T *p = std::get_if<T>(&attr);
        POWERSERVE_ASSERT(p != nullptr, "op carries no attributes of the requested kind");
        return *p;
    }
    TensorNode *input(size_t i) const { POWERSERVE_ASSERT(i < in.size()); return in[i]; }
    TensorNode *output() const { POWERSERVE_ASSERT(out.size() == 1); return out[0]; }
};

struct Graph {
    std::vector<std::shared_ptr<TensorNode>> tensors; // everything the graph mentions, in creation order
    std::vector<std::shared_ptr<OpNode>> ops;         // program order
    std::string m_model_id;
    explicit Graph(std::string model_id) : m_model_id(std::move(model_id)) {}

    // ---- tensors
    auto add_tensor(const Tensor &t) -> TensorNode *;                 // an existing tensor (weight, cache): shares its buffer
    auto new_tensor(DataType dt, const Shape &s) -> TensorNode *;     // an intermediate the executor allocates
    auto dup_tensor(TensorNode *t) -> TensorNode * { return new_tensor(t->m_dtype, t->m_shape); }
    auto view_tensor(const TensorNode *t, Shape shape) -> TensorViewNode *; // same storage, new shape, no op

    // ---- ops (reference builder API)
    auto get_embedding(TensorNode *weight, const std::vector<int> &tokens) -> TensorNode *;
    auto add(TensorNode *a, TensorNode *b) -> TensorNode *;
    auto mat_mul(TensorNode *a, TensorNode *b) -> TensorNode *;
    auto rms_norm(TensorNode *x, TensorNode *weight, float eps) -> TensorNode *;
    auto silu_hadamard(TensorNode *gate, TensorNode *up) -> TensorNode *;
    void copy(TensorNode *dst, TensorNode *src);
    auto rope(TensorNode *src, const std::vector<int> &pos, const ModelConfig::LLMConfig::RopeConfig &params) -> TensorNode *;
    auto softmax(TensorNode *x) -> TensorNode *; // src/graph/graph.cpp:118-125
    auto softmax_ext(TensorNode *x, TensorNode *mask, float scale, float max_bias) -> TensorNode *;
    auto permute(TensorNode *x, Shape axes) -> TensorViewNode *;
    auto cont(TensorNode *x, Shape shape) -> TensorNode *;
    auto view(const TensorNode *x, Shape shape, Shape stride, size_t offset = 0) -> TensorViewNode *;
    auto get_mask(const CausalAttentionMask &mask, Shape shape, const std::vector<int> &pos) -> TensorNode *;
    auto transpose(TensorNode *x) -> TensorViewNode *;

private:
    // appends one op; `result` (may be null: COPY writes into its first input) becomes its single output
    auto emit(OpType kind, std::initializer_list<TensorNode *> inputs, TensorNode *result, OpAttr attr = {}) -> TensorNode *;
};

} // namespace powerserve
