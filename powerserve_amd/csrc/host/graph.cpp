// Shape rules follow src/graph/graph.cpp:20-266 (e.g. mat_mul(a, b): a.shape[0] == b.shape[0], result
// {a.shape[1], b.shape[1], b.shape[2], b.shape[3]} FP32; copy(dst, src) has dst first and no output).
#include "graph.hpp"

namespace powerserve {

auto Graph::get_embedding(TensorNode *weight, const std::vector<int> &tokens) -> TensorNode * {
    auto out = dup_tensor(weight);
    out->m_dtype = DataType::FP32;
    out->m_shape[1] = tokens.size();
    auto op = new_op(OpType::GET_EMBEDDING);
    op->set_inputs({weight}); op->set_outputs({out}); op->set_params(GetEmbeddingParams{tokens});
    return out;
}
auto Graph::add(TensorNode *a, TensorNode *b) -> TensorNode * {
    POWERSERVE_ASSERT(tensor_can_repeat(b, a));
    auto out = dup_tensor(a);
    auto op = new_op(OpType::ADD);
    op->set_inputs({a, b}); op->set_outputs({out});
    return out;
}
auto Graph::mat_mul(TensorNode *a, TensorNode *b) -> TensorNode * {
    POWERSERVE_ASSERT(a->m_shape[0] == b->m_shape[0]);
    POWERSERVE_ASSERT(tensor_can_mul_mat(a, b));
    auto out = new_tensor(DataType::FP32, {a->m_shape[1], b->m_shape[1], b->m_shape[2], b->m_shape[3]});
    auto op = new_op(OpType::MAT_MUL);
    op->set_inputs({a, b}); op->set_outputs({out});
    return out;
}
auto Graph::rms_norm(TensorNode *x, TensorNode *weight, float eps) -> TensorNode * {
    POWERSERVE_ASSERT(weight->n_dims() == 1);
    POWERSERVE_ASSERT(x->m_dtype == weight->m_dtype);
    POWERSERVE_ASSERT(x->m_shape[0] == weight->m_shape[0]);
    auto out = dup_tensor(x);
    auto op = new_op(OpType::RMS_NORM);
    op->set_inputs({x, weight}); op->set_outputs({out}); op->set_params(RMSNormParams{eps});
    return out;
}
auto Graph::silu_hadamard(TensorNode *gate, TensorNode *up) -> TensorNode * {
    POWERSERVE_ASSERT(gate->m_dtype == up->m_dtype);
    POWERSERVE_ASSERT(gate->m_shape == up->m_shape);
    auto out = dup_tensor(gate);
    auto op = new_op(OpType::SILU_HADAMARD);
    op->set_inputs({gate, up}); op->set_outputs({out});
    return out;
}
void Graph::copy(TensorNode *dst, TensorNode *src) {
    auto op = new_op(OpType::COPY);
    op->set_inputs({dst, src}); op->set_params(CopyParams{});
}
auto Graph::rope(TensorNode *src, const std::vector<int> &pos, const ModelConfig::LLMConfig::RopeConfig &params) -> TensorNode * {
    auto out = dup_tensor(src);
    auto op = new_op(OpType::ROPE);
    op->set_inputs({src}); op->set_outputs({out}); op->set_params(RopeParams{pos, params});
    return out;
}
auto Graph::softmax_ext(TensorNode *x, TensorNode *mask, float scale, float max_bias) -> TensorNode * {
    auto out = dup_tensor(x);
    auto op = new_op(OpType::SOFTMAX_EXT);
    op->set_inputs({x, mask}); op->set_outputs({out}); op->set_params(SoftmaxExtParams{scale, max_bias});
    return out;
}
auto Graph::permute(TensorNode *x, Shape axes) -> TensorViewNode * {
    for (int i = 0; i < 4; i++) { POWERSERVE_ASSERT(axes[i] < max_n_dims); for (int j = i + 1; j < 4; j++) POWERSERVE_ASSERT(axes[i] != axes[j]); }
    Shape shape{};
    for (int i = 0; i < 4; i++) shape[axes[i]] = x->m_shape[i];
    auto out = view_tensor(x, shape);
    auto op = new_op(OpType::PERMUTE);
    op->set_inputs({x}); op->set_outputs({out}); op->set_params(PermuteParams{axes});
    return out;
}
auto Graph::cont(TensorNode *x, Shape shape) -> TensorNode * {
    auto out = new_tensor(x->m_dtype, shape);
    auto op = new_op(OpType::CONT);
    op->set_inputs({x}); op->set_outputs({out}); op->set_params(ContParams{});
    return out;
}
auto Graph::view(const TensorNode *x, Shape shape, Shape stride, size_t offset) -> TensorViewNode * {
    auto out = view_tensor(x, shape);
    auto op = new_op(OpType::VIEW);
    op->set_inputs({}); op->set_outputs({out}); op->set_params(ViewParams{stride, offset});
    return out;
}
auto Graph::get_mask(const CausalAttentionMask &mask, Shape shape, const std::vector<int> &pos) -> TensorNode * {
    auto out = new_tensor(DataType::FP32, shape);
    auto op = new_op(OpType::GET_MASK);
    op->set_outputs({out}); op->set_params(GetMaskParams{mask, pos});
    return out;
}
auto Graph::transpose(TensorNode *x) -> TensorViewNode * {
    auto shape = x->m_shape;
    std::swap(shape[0], shape[1]);
    auto out = view_tensor(x, shape);
    auto op = new_op(OpType::TRANSPOSE);
    op->set_inputs({x}); op->set_outputs({out});
    return out;
}

} // namespace powerserve
