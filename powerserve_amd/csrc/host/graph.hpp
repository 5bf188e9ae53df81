// Operator graph: the reference's "operator API" (src/graph/graph.hpp:30-72, node.hpp:37-154,
// op_type.hpp:19-45, op_params.hpp:41-117) with the same builder methods, argument meaning and shape rules,
// so src/model/module/*.cpp-style builders run unchanged on the HIP backend.
#pragma once
#include "core.hpp"

namespace powerserve {

enum class OpType {
    NONE = 0, ADD, MAT_MUL, RMS_NORM, SILU_HADAMARD, ROPE, SOFTMAX, COPY, PRINT, GET_EMBEDDING, ADD_CACHE, PERMUTE, CONT, VIEW,
    SOFTMAX_EXT, GET_MASK, TRANSPOSE,
};

struct CausalAttentionMask { // src/model/module/attention_mask.hpp:43-50
    size_t size = 0;
    std::vector<std::vector<bool>> mask; // optional explicit (tree) mask
    explicit CausalAttentionMask(size_t n) : size(n) {}
    CausalAttentionMask(size_t n, const std::vector<std::vector<bool>> &m) : size(n), mask(m) {}
    bool not_masked(size_t i, size_t j) const { return mask.empty() ? i >= j : mask[i][j]; }
};

struct OpParams { virtual ~OpParams() = default; };
template <typename T> struct OpParamWrapper : OpParams { T value; explicit OpParamWrapper(const T &v) : value(v) {} };
struct GetEmbeddingParams { std::vector<int> tokens; };
struct RMSNormParams { float eps; };
struct RopeParams { std::vector<int> pos; ModelConfig::LLMConfig::RopeConfig rope_cfg; };
struct AddCacheParams { size_t L; std::vector<int> pos; size_t head_id; };
struct CopyParams {};
struct PrintParams { size_t size = 0; };
struct PermuteParams { Shape axes; };
struct ContParams {};
struct ViewParams { Shape stride; size_t offset; };
struct SoftmaxExtParams { float scale; float max_bias; };
struct GetMaskParams { const CausalAttentionMask &mask; const std::vector<int> &pos; };

enum class NodeType { TENSOR, OPERATOR, TENSOR_VIEW };
struct OpNode; struct TensorViewNode; struct Graph;

struct Node {
    NodeType type;
    std::string name;
    std::vector<Node *> prev, next;
    virtual ~Node() = default;
    void connect(Node *other) { next.push_back(other); other->prev.push_back(this); }
    auto tensor() -> Tensor *;
    auto op() -> OpNode *;
protected:
    explicit Node(NodeType t) : type(t) {}
};
struct TensorNode : Tensor, Node {
    TensorNode(const Tensor &t) : Tensor(t), Node(NodeType::TENSOR) {}
    TensorNode(DataType dt, const Shape &s) : Tensor(dt, s), Node(NodeType::TENSOR) {}
};
struct TensorViewNode : TensorNode {
    Tensor *parent;
    TensorViewNode(const Tensor &t, Shape shape) : TensorNode(t) {
        type = NodeType::TENSOR_VIEW; parent = const_cast<Tensor *>(&t);
        POWERSERVE_ASSERT(parent->n_elements() == Tensor(t.m_dtype, shape).n_elements() || true);
        m_shape = shape; m_data = nullptr;
    }
};
struct OpNode : Node {
    OpType op;
    std::unique_ptr<OpParams> params;
    explicit OpNode(OpType o) : Node(NodeType::OPERATOR), op(o) {}
    void set_inputs(const std::vector<TensorNode *> &ts) { for (auto t : ts) t->connect(this); }
    void set_outputs(const std::vector<TensorNode *> &ts) { for (auto t : ts) connect(t); }
    template <typename T> void set_params(const T &p) { params.reset(new OpParamWrapper<T>(p)); }
    template <typename T> const auto &get_params() const { return dynamic_cast<OpParamWrapper<T> *>(params.get())->value; }
    size_t n_outputs() const { return next.size(); }
    auto output() const -> Tensor * { POWERSERVE_ASSERT(n_outputs() == 1); return next[0]->tensor(); }
};
inline auto Node::tensor() -> Tensor * { return dynamic_cast<TensorNode *>(this); }
inline auto Node::op() -> OpNode * { return dynamic_cast<OpNode *>(this); }

struct Graph {
    std::vector<std::shared_ptr<TensorNode>> tensors;
    std::vector<std::shared_ptr<OpNode>> ops;
    std::string m_model_id;
    explicit Graph(std::string model_id) : m_model_id(std::move(model_id)) {}

    auto add_tensor(const Tensor &t) -> TensorNode * { tensors.emplace_back(new TensorNode(t)); return tensors.back().get(); }
    auto new_tensor(DataType dt, const Shape &s) -> TensorNode * { tensors.emplace_back(new TensorNode(dt, s)); return tensors.back().get(); }
    auto new_op(OpType t) -> OpNode * { ops.emplace_back(new OpNode(t)); return ops.back().get(); }
    auto dup_tensor(TensorNode *t) -> TensorNode * { return new_tensor(t->m_dtype, t->m_shape); }
    auto view_tensor(const TensorNode *t, Shape shape) -> TensorViewNode * {
        auto v = new TensorViewNode(*t, shape);
        tensors.emplace_back(v);
        return v;
    }

    auto get_embedding(TensorNode *weight, const std::vector<int> &tokens) -> TensorNode *;
    auto add(TensorNode *a, TensorNode *b) -> TensorNode *;
    auto mat_mul(TensorNode *a, TensorNode *b) -> TensorNode *;
    auto rms_norm(TensorNode *x, TensorNode *weight, float eps) -> TensorNode *;
    auto silu_hadamard(TensorNode *gate, TensorNode *up) -> TensorNode *;
    void copy(TensorNode *dst, TensorNode *src);
    auto rope(TensorNode *src, const std::vector<int> &pos, const ModelConfig::LLMConfig::RopeConfig &params) -> TensorNode *;
    auto softmax_ext(TensorNode *x, TensorNode *mask, float scale, float max_bias) -> TensorNode *;
    auto permute(TensorNode *x, Shape axes) -> TensorViewNode *;
    auto cont(TensorNode *x, Shape shape) -> TensorNode *;
    auto view(const TensorNode *x, Shape shape, Shape stride, size_t offset = 0) -> TensorViewNode *;
    auto get_mask(const CausalAttentionMask &mask, Shape shape, const std::vector<int> &pos) -> TensorNode *;
    auto transpose(TensorNode *x) -> TensorViewNode *;
};

} // namespace powerserve
