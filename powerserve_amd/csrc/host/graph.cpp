#include "graph.hpp"

namespace powerserve {

auto Graph::add_tensor(const Tensor &t) -> TensorNode * {
    tensors.push_back(std::make_shared<TensorNode>(t));
    return tensors.back().get();
}
auto Graph::new_tensor(DataType dt, const Shape &s) -> TensorNode * {
    tensors.push_back(std::make_shared<TensorNode>(dt, s));
    return tensors.back().get();
}
auto Graph::view_tensor(const TensorNode *t, Shape shape) -> TensorViewNode * {
    TensorNode *v = add_tensor(*t);
    v->m_shape  = shape;
    v->m_data   = nullptr; // bound to the source's storage when buffers are assigned
    v->alias_of = t;
    return v;
}

auto Graph::emit(OpType kind, std::initializer_list<TensorNode *> inputs, TensorNode *result, OpAttr attr) -> TensorNode * {
    auto op = std::make_shared<OpNode>(kind);
    op->in.assign(inputs.begin(), inputs.end());
    op->attr = std::move(attr);
    if (result) {
        op->out.push_back(result);
        result->producer = op.get();
    }
    ops.push_back(std::move(op));
    return result;
}

// ---- shape inference.  x keeps its shape through the elementwise / normalising ops; the rules that are not "same as
// the first input" are spelled out where they apply.

auto Graph::get_embedding(TensorNode *weight, const std::vector<int> &tokens) -> TensorNode * {
    Shape rows = weight->m_shape; // one table row per token, widened to FP32
    rows[1] = tokens.size();
    return emit(OpType::GET_EMBEDDING, {weight}, new_tensor(DataType::FP32, rows), GetEmbeddingParams{tokens});
}

auto Graph::add(TensorNode *a, TensorNode *b) -> TensorNode * {
    POWERSERVE_ASSERT(tensor_can_repeat(b, a)); // b broadcasts over a
    return emit(OpType::ADD, {a, b}, dup_tensor(a));
}

auto Graph::mat_mul(TensorNode *a, TensorNode *b) -> TensorNode * {
    // a: [K, M, ...] (weights / keys), b: [K, N, ...] (activations)  ->  [M, N, b.2, b.3] FP32
    POWERSERVE_ASSERT(a->m_shape[0] == b->m_shape[0]);
    POWERSERVE_ASSERT(tensor_can_mul_mat(a, b));
    return emit(OpType::MAT_MUL, {a, b}, new_tensor(DataType::FP32, {a->m_shape[1], b->m_shape[1], b->m_shape[2], b->m_shape[3]}));
}

auto Graph::rms_norm(TensorNode *x, TensorNode *weight, float eps) -> TensorNode * {
    POWERSERVE_ASSERT(weight->n_dims() == 1 && weight->m_shape[0] == x->m_shape[0]);
    POWERSERVE_ASSERT(x->m_dtype == weight->m_dtype);
    return emit(OpType::RMS_NORM, {x, weight}, dup_tensor(x), RMSNormParams{eps});
}

auto Graph::silu_hadamard(TensorNode *gate, TensorNode *up) -> TensorNode * {
    POWERSERVE_ASSERT(gate->m_dtype == up->m_dtype && gate->m_shape == up->m_shape);
    return emit(OpType::SILU_HADAMARD, {gate, up}, dup_tensor(gate));
}

void Graph::copy(TensorNode *dst, TensorNode *src) { emit(OpType::COPY, {dst, src}, nullptr); } // destination first, no result

auto Graph::rope(TensorNode *src, const std::vector<int> &pos, const ModelConfig::LLMConfig::RopeConfig &params) -> TensorNode * {
    return emit(OpType::ROPE, {src}, dup_tensor(src), RopeParams{pos, params});
}

auto Graph::softmax(TensorNode *x) -> TensorNode * { return emit(OpType::SOFTMAX, {x}, dup_tensor(x)); }

auto Graph::softmax_ext(TensorNode *x, TensorNode *mask, float scale, float max_bias) -> TensorNode * {
    return emit(OpType::SOFTMAX_EXT, {x, mask}, dup_tensor(x), SoftmaxExtParams{scale, max_bias});
}

auto Graph::permute(TensorNode *x, Shape axes) -> TensorViewNode * {
    Shape shape{};
    unsigned used = 0; // axes must be a permutation of 0..3: dimension i of x becomes dimension axes[i]
    for (size_t i = 0; i < max_n_dims; i++) {
        POWERSERVE_ASSERT(axes[i] < max_n_dims && !(used >> axes[i] & 1u));
        used |= 1u << axes[i];
        shape[axes[i]] = x->m_shape[i];
    }
    return emit(OpType::PERMUTE, {x}, view_tensor(x, shape), PermuteParams{axes});
}

auto Graph::cont(TensorNode *x, Shape shape) -> TensorNode * { return emit(OpType::CONT, {x}, new_tensor(x->m_dtype, shape)); }

auto Graph::view(const TensorNode *x, Shape shape, Shape stride, size_t offset) -> TensorViewNode * {
    return emit(OpType::VIEW, {}, view_tensor(x, shape), ViewParams{stride, offset}); // x is reached through alias_of
}

auto Graph::get_mask(const CausalAttentionMask &mask, Shape shape, const std::vector<int> &pos) -> TensorNode * {
    return emit(OpType::GET_MASK, {}, new_tensor(DataType::FP32, shape), GetMaskParams{&mask, pos});
}

auto Graph::transpose(TensorNode *x) -> TensorViewNode * {
    Shape shape = x->m_shape;
    std::swap(shape[0], shape[1]);
    return emit(OpType::TRANSPOSE, {x}, view_tensor(x, shape));
}

} // namespace powerserve
