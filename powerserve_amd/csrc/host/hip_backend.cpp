#include "hip_backend.hpp"

#include <cmath>
#include <cstring>

namespace powerserve {
namespace hip {

// ---------------------------------------------------------------- KV
HIPKV::HIPKV(const ModelConfig::LLMConfig &c, ps_hip_model *model) :
    m_kv_dim(c.kv_dim), m_n_kv_heads(c.n_kv_heads), m_n_ctx(c.seq_len), m_n_layers(c.n_layers), m_head_size(c.head_size), m_model(model) {
    Stride stride = {sizeof(float), sizeof(float) * m_n_ctx, sizeof(float) * m_kv_dim * m_n_ctx, sizeof(float) * m_kv_dim * m_n_ctx};
    for (size_t L = 0; L < m_n_layers; L++) {
        key_tensors.emplace_back(Tensor(DataType::FP32, {m_n_ctx, m_kv_dim, 1, 1}));
        value_tensors.emplace_back(Tensor(DataType::FP32, {m_n_ctx, m_kv_dim, 1, 1}));
        key_tensors[L].m_data   = std::make_shared<HIPBuffer>(stride, (void *)ps_hip_model_k_cache(model, (int)L));
        value_tensors[L].m_data = std::make_shared<HIPBuffer>(stride, (void *)ps_hip_model_v_cache(model, (int)L));
    }
}
void HIPKV::advance(int n) {
    POWERSERVE_ASSERT(position() + (size_t)n <= m_n_ctx, "the length of kvcache is up to the preset threshold");
    POWERSERVE_ASSERT(ps_hip_model_kv_advance(m_model, (size_t)n) == 0, "kv advance refused: a pending lowered forward has no valid result");
}
void HIPKV::rollback(size_t n) {
    POWERSERVE_ASSERT(position() >= n);
    ps_hip_model_kv_rollback(m_model, n);
}
void HIPKV::copy(size_t dst, size_t src_token) { POWERSERVE_ASSERT(ps_hip_model_kv_copy(m_model, dst, src_token) == 0, "kv copy: index out of range"); }
void HIPKV::move(size_t dst, size_t src) { POWERSERVE_ASSERT(ps_hip_model_kv_move(m_model, dst, src) == 0, "kv move: index out of range"); }
void HIPKV::mask(size_t i) { POWERSERVE_ASSERT(i < position()); ps_hip_model_kv_mask(m_model, i, 0); }     // kv_cache.hpp:211-214
void HIPKV::unmask(size_t i) { POWERSERVE_ASSERT(i < position()); ps_hip_model_kv_mask(m_model, i, 1); }   // kv_cache.hpp:216-219
void HIPKV::save_tokens(size_t n) { POWERSERVE_ASSERT(ps_hip_model_kv_save_tokens(m_model, n) == 0, "the length of kvcache is up to the preset threshold"); }
void HIPKV::unmask_tokens(size_t n) { POWERSERVE_ASSERT(ps_hip_model_kv_unmask_tokens(m_model, n) == 0, "the length of kvcache is up to the preset threshold"); }
size_t HIPKV::append_tokens(size_t n) {
    size_t old = 0;
    POWERSERVE_ASSERT(ps_hip_model_kv_append_tokens(m_model, n, &old) == 0, "kv append refused");
    return old;
}

// ---------------------------------------------------------------- backend
HIPBackend::HIPBackend(const ModelConfig::LLMConfig &config, const HyperParams &, int device) : m_config(config), m_device(device) {
    if (ps_hip_create(device, &m_ctx) != 0) POWERSERVE_ABORT("ps_hip_create failed: no usable HIP device (there is no CPU fallback)");
}
HIPBackend::~HIPBackend() {
    if (m_arena) ps_hip_free(m_ctx, m_arena);
    m_kv.reset();
    if (m_model) ps_hip_model_destroy(m_model);
    if (m_ctx) ps_hip_destroy(m_ctx);
}
void HIPBackend::attach_model(ps_hip_model *m, LoweringTable table) {
    m_model = m;
    m_kv    = std::make_unique<HIPKV>(m_config, m);
    m_table = std::move(table);
}
void HIPBackend::check(int rc, const char *what) const {
    if (rc != 0) POWERSERVE_ABORT(std::string(what) + ": " + ps_hip_last_error(m_ctx)); // C-ABI error -> reference abort/throw
}
void HIPBackend::sync() const { check(ps_hip_sync(m_ctx), "sync"); }
void HIPBackend::reset_threadpool() { sync(); }

void HIPBackend::arena_reserve(size_t bytes) {
    if (bytes <= m_arena_cap) return;
    sync();
    if (m_arena) check(ps_hip_free(m_ctx, m_arena), "arena free");
    size_t cap = std::max(bytes, m_arena_cap * 2);
    void *p = nullptr;
    check(ps_hip_malloc(m_ctx, cap, &p), "arena malloc");
    m_arena = (char *)p; m_arena_cap = cap; m_arena_off = 0;
}
void *HIPBackend::arena_alloc(size_t bytes) {
    bytes = (bytes + 255) / 256 * 256;
    if (m_arena_off + bytes > m_arena_cap) POWERSERVE_ABORT("arena exhausted: Executor::allocate_buffers must reserve first");
    void *p = m_arena + m_arena_off;
    m_arena_off += bytes;
    return p;
}

ps_tensor HIPBackend::to_ps(const Tensor *t) const { // convert_to_ggml (src/backend/ggml/ggml.hpp:87-96)
    ps_tensor p{};
    p.dtype = to_ggml_type(t->m_dtype);
    auto &b = t->get<HIPBuffer>();
    p.data  = b.m_data;
    for (size_t i = 0; i < max_n_dims; i++) { p.ne[i] = (int64_t)t->m_shape[i]; p.nb[i] = b.m_stride[i]; }
    return p;
}

void HIPBackend::matmul(const Tensor *dst, const Tensor *src0, const Tensor *src1) const {
    auto d = to_ps(dst), a = to_ps(src0), b = to_ps(src1);
    check(ps_hip_mul_mat(m_ctx, &d, &a, &b), "matmul");
}
void HIPBackend::rmsnorm(const Tensor *out, const Tensor *x, const Tensor *weight, float eps) const {
    auto d = to_ps(out), a = to_ps(x), w = to_ps(weight);
    check(ps_hip_rms_norm(m_ctx, &d, &a, &w, eps), "rmsnorm");
}
int HIPBackend::get_n_tasks(std::shared_ptr<OpNode>) { return 1; }

void HIPBackend::add_cache(const Tensor *k, const Tensor *v, size_t L, const std::vector<int> &pos, size_t) {
    const size_t bs = pos.size(), kvd = m_kv->m_kv_dim, cur = m_kv->position();
    POWERSERVE_ASSERT(bs == m_kv->m_batch_size && L < m_kv->m_n_layers && cur + bs <= m_kv->m_n_ctx);
    auto [kc, vc] = m_kv->get_cache(L);
    // K rows are contiguous behind the position; V is kept transposed ([kv_dim][n_ctx]): a strided view of the cache as the destination
    Tensor kd(DataType::FP32, {kvd, bs, 1, 1}), vd(DataType::FP32, {kvd, bs, 1, 1});
    kd.m_data = std::make_shared<HIPBuffer>(Stride{4, 4 * kvd, 4 * kvd * bs, 4 * kvd * bs}, (char *)kc.get<HIPBuffer>().m_data + cur * kvd * 4);
    vd.m_data = std::make_shared<HIPBuffer>(Stride{4 * m_kv->m_n_ctx, 4, 4 * kvd * m_kv->m_n_ctx, 4 * kvd * m_kv->m_n_ctx}, (char *)vc.get<HIPBuffer>().m_data + cur * 4);
    copy(&kd, k);
    copy(&vd, v);
}

void HIPBackend::softmax(const Tensor *out, const Tensor *x) const {
    auto d = to_ps(out), a = to_ps(x);
    check(ps_hip_soft_max(m_ctx, &d, &a), "softmax");
}
void HIPBackend::rope(Tensor *out, const Tensor *src, const std::vector<int> &pos, const ModelConfig::LLMConfig::RopeConfig &c) const {
    auto d = to_ps(out), a = to_ps(src);
    ps_rope_params rp{c.n_dims, c.n_ctx_orig, c.freq_base, c.freq_scale, c.ext_factor, c.attn_factor, c.beta_fast, c.beta_slow, c.rope_type};
    std::vector<int32_t> p(pos.begin(), pos.end());
    check(ps_hip_rope(m_ctx, &d, &a, p.data(), (int)p.size(), &rp), "rope");
}
void HIPBackend::add(const Tensor *dst, const Tensor *src0, const Tensor *src1) const {
    auto d = to_ps(dst), a = to_ps(src0), b = to_ps(src1);
    check(ps_hip_add(m_ctx, &d, &a, &b), "add");
}
void HIPBackend::permute(const Tensor *out, const Tensor *x, Shape axes) const { // metadata only (ggml_wrapper.cpp:125-133)
    Stride stride{};
    for (int i = 0; i < 4; i++) stride[axes[i]] = x->get<HIPBuffer>().m_stride[i];
    out->get<HIPBuffer>().m_stride = stride;
}
void HIPBackend::transpose(const Tensor *out, const Tensor *x) const { // metadata only (ggml.cpp:170-177)
    Stride stride{x->get<HIPBuffer>().m_stride};
    std::swap(stride[0], stride[1]);
    out->get<HIPBuffer>().m_data   = x->get<HIPBuffer>().m_data;
    out->get<HIPBuffer>().m_stride = stride;
}
void HIPBackend::cont(const Tensor *out, const Tensor *x) const {
    auto d = to_ps(out), a = to_ps(x);
    check(ps_hip_dup(m_ctx, &d, &a), "cont");
}
void HIPBackend::copy(const Tensor *dst, const Tensor *src) const {
    auto d = to_ps(dst), a = to_ps(src);
    check(ps_hip_dup(m_ctx, &d, &a), "copy");
}
void HIPBackend::softmax_ext(const Tensor *out, const Tensor *x, const Tensor *mask, float scale, float max_bias) const {
    auto d = to_ps(out), a = to_ps(x), m = to_ps(mask);
    check(ps_hip_softmax_ext(m_ctx, &d, &a, &m, scale, max_bias), "softmax_ext");
}
void HIPBackend::silu_hadamard(const Tensor *out, const Tensor *hb, const Tensor *hb2) const {
    POWERSERVE_ASSERT(is_contiguous(out, 0) && is_contiguous(hb, 0) && is_contiguous(hb2, 0));
    auto d = to_ps(out), a = to_ps(hb), b = to_ps(hb2);
    check(ps_hip_silu_hadamard(m_ctx, &d, &a, &b), "silu_hadamard");
}
void HIPBackend::get_embedding(const Tensor *dst, const Tensor *weight, const std::vector<int> &tokens) const {
    POWERSERVE_ASSERT(tokens.size() == dst->m_shape[1]);
    auto d = to_ps(dst), w = to_ps(weight);
    std::vector<int32_t> t(tokens.begin(), tokens.end());
    check(ps_hip_get_embedding(m_ctx, &d, &w, t.data(), (int)t.size()), "get_embedding");
}
void HIPBackend::get_mask(const Tensor *out, const std::vector<int> &pos, const CausalAttentionMask &mask) const {
    auto d = to_ps(out);
    std::vector<int32_t> p(pos.begin(), pos.end());
    std::vector<uint8_t> tree;
    if (!mask.mask.empty()) {
        tree.resize(pos.size() * pos.size());
        for (size_t i = 0; i < pos.size(); i++) for (size_t j = 0; j < pos.size(); j++) tree[i * pos.size() + j] = mask.mask[i][j] ? 1 : 0;
    }
    check(ps_hip_get_mask(m_ctx, &d, p.data(), (int)p.size(), tree.empty() ? nullptr : tree.data()), "get_mask");
}
bool HIPBackend::is_contiguous(const Tensor *t, int n) const { // ggml_is_contiguous_n
    auto &s = t->get<HIPBuffer>().m_stride;
    size_t next = get_type_size(t->m_dtype);
    if (t->m_shape[0] != get_block_size(t->m_dtype) && s[0] != next) return false;
    next *= t->m_shape[0] / get_block_size(t->m_dtype);
    for (int i = 1; i < 4; i++) {
        if (t->m_shape[i] != 1) { if (i > n) { if (s[i] != next) return false; next *= t->m_shape[i]; } else next = t->m_shape[i] * s[i]; }
    }
    return true;
}
int HIPBackend::get_vec_dot_type(const Tensor *t) const { return ps_hip_vec_dot_type(to_ggml_type(t->m_dtype)); }
void HIPBackend::print(const Tensor *x, size_t) const {
    POWERSERVE_ASSERT(x->m_dtype == DataType::FP32);
    std::vector<float> h(x->n_elements());
    sync();
    check(ps_hip_memcpy_d2h(m_ctx, h.data(), x->get<HIPBuffer>().m_data, h.size() * 4), "print");
    for (float v : h) std::printf("%.6f\n", (double)v);
}
// ---- plan(): recognise the canonical forward and lower it
namespace {
const void *handle_of(const TensorNode *t) { // device handle behind a graph tensor (weights: ps_weight*, F32 vectors: device float*)
    return (t && t->m_data) ? t->get<HIPBuffer>().m_data : nullptr;
}
struct Cursor {
    std::vector<std::shared_ptr<OpNode>> &ops;
    size_t i = 0;
    OpNode *take(OpType t) { return (i < ops.size() && ops[i]->op == t) ? ops[i++].get() : nullptr; }
    bool done() const { return i == ops.size(); }
};
} // namespace

bool HIPBackend::match_canonical(std::vector<std::shared_ptr<OpNode>> &ops, Lowered &out) const {
    const auto &T = m_table;
    if (!m_model || T.layers.size() != m_config.n_layers || T.layers.empty()) return false;
    Cursor c{ops};
    // x = get_embedding(token table, tokens)
    OpNode *emb = c.take(OpType::GET_EMBEDDING);
    if (!emb || handle_of(emb->in[0]) != T.token_embd) return false;
    const auto &tokens = emb->get_params<GetEmbeddingParams>().tokens;
    const size_t bs = tokens.size();
    if (bs == 0) return false;
    TensorNode *x = emb->out[0];
    const std::vector<int> *pos = nullptr;
    const CausalAttentionMask *mask = nullptr;
    const float want_scale = 1.0f / sqrtf((float)m_config.head_size);
    const size_t hs = m_config.head_size, n_head = m_config.n_heads, n_head_kv = m_config.n_kv_heads, n_ctx = m_config.seq_len, kv_gqa = hs * n_head_kv, es = sizeof(float);
    if (bs > (size_t)ps_hip_model_max_batch(m_model)) return false; // (run op by op: the lowered forward's buffers hold max_batch columns)
    auto mat_mul = [&](const void *w, TensorNode *act) -> TensorNode * { // MAT_MUL(weight w, activation act) -> its output node
        OpNode *o = c.take(OpType::MAT_MUL);
        return (o && handle_of(o->in[0]) == w && o->in[1] == act) ? o->out[0] : nullptr;
    };
    auto rms = [&](const void *w, TensorNode *in) -> TensorNode * {
        OpNode *o = c.take(OpType::RMS_NORM);
        return (o && o->in[0] == in && handle_of(o->in[1]) == w && o->get_params<RMSNormParams>().eps == m_config.norm_eps) ? o->out[0] : nullptr;
    };
    auto add_bias = [&](TensorNode *in, const void *b) -> TensorNode * {
        if (!T.bias) return in;
        OpNode *o = c.take(OpType::ADD);
        return (o && o->in[0] == in && handle_of(o->in[1]) == b) ? o->out[0] : nullptr;
    };
    for (size_t L = 0; L < T.layers.size(); L++) {
        const auto &W = T.layers[L];
        // ---- NormAttention::build
        TensorNode *n1 = rms(W.attn_norm, x);
        if (!n1) return false;
        TensorNode *q = mat_mul(W.wq, n1); if (!q || !(q = add_bias(q, W.bq))) return false;
        TensorNode *k = mat_mul(W.wk, n1); if (!k || !(k = add_bias(k, W.bk))) return false;
        TensorNode *v = mat_mul(W.wv, n1); if (!v || !(v = add_bias(v, W.bv))) return false;
        TensorNode *roped[2] = {nullptr, nullptr};
        for (int r = 0; r < 2; r++) { // rope(q view), rope(k view): views of q / k split into heads, same positions, the model's rope configuration
            OpNode *o = c.take(OpType::ROPE);
            if (!o) return false;
            const auto &rp = o->get_params<RopeParams>();
            if (!pos) pos = &rp.pos;
            if (rp.pos != *pos || rp.pos.size() != bs || rp.rope_cfg.n_dims != m_config.rope_config.n_dims || rp.rope_cfg.rope_type != m_config.rope_config.rope_type ||
                rp.rope_cfg.freq_base != m_config.rope_config.freq_base || rp.rope_cfg.freq_scale != m_config.rope_config.freq_scale)
                return false;
            const TensorNode *src = o->in[0];
            if (!src || src->alias_of != (r == 0 ? q : k) || src->m_shape != Shape{hs, r == 0 ? n_head : n_head_kv, bs, 1}) return false;
            roped[r] = o->out[0];
        }
        // The attention core is lowered to the fused kernels, which hard-wire what NormAttention::build wires: every operand,
        // view offset / stride, permutation and n_kv is compared — a graph that keeps the op order but reads another window of
        // the cache, another n_kv or other strides runs op by op instead of being silently replaced by the stock forward.
        const size_t cur_pos = (size_t)(*pos)[0], n_kv = (size_t)pos->back() + 1;
        const void *kc_dev = ps_hip_model_k_cache(m_model, (int)L), *vc_dev = ps_hip_model_v_cache(m_model, (int)L);
        auto view_of = [&](const void *cache, const Shape &shape, const Shape &stride, size_t offset) -> TensorNode * {
            OpNode *o = c.take(OpType::VIEW);
            if (!o) return nullptr;
            const TensorNode *t = o->out[0];
            const auto &vp = o->get_params<ViewParams>();
            return (t->alias_of && handle_of(t->alias_of) == cache && t->m_shape == shape && vp.stride == stride && vp.offset == offset) ? o->out[0] : nullptr;
        };
        // KV store: transpose(v), view(k cache rows cur_pos ..), copy, view(v cache column cur_pos ..), copy
        OpNode *tr = c.take(OpType::TRANSPOSE);
        if (!tr || tr->in[0] != v) return false;
        TensorNode *kc = view_of(kc_dev, Shape{bs * kv_gqa, 1, 1, 1}, Shape{es, es * bs * kv_gqa, es * bs * kv_gqa, es * bs * kv_gqa}, es * kv_gqa * cur_pos);
        OpNode *cpk = kc ? c.take(OpType::COPY) : nullptr;
        if (!cpk || cpk->in[0] != kc || cpk->in[1] != roped[1]) return false;
        TensorNode *vc = view_of(vc_dev, Shape{bs, kv_gqa, 1, 1}, Shape{es, n_ctx * es, n_ctx * es * kv_gqa, n_ctx * es * kv_gqa}, es * cur_pos);
        OpNode *cpv = vc ? c.take(OpType::COPY) : nullptr;
        if (!cpv || cpv->in[0] != vc || cpv->in[1] != tr->out[0]) return false;
        // scores, mask, softmax, V.p, merge heads
        const Shape heads_perm{0, 2, 1, 3};
        OpNode *qp = c.take(OpType::PERMUTE);
        if (!qp || qp->in[0] != roped[0] || qp->get_params<PermuteParams>().axes != heads_perm) return false;
        TensorNode *kv = view_of(kc_dev, Shape{hs, n_kv, n_head_kv, 1}, Shape{es, es * kv_gqa, es * hs, es * hs * n_head_kv}, 0);
        OpNode *kq = kv ? c.take(OpType::MAT_MUL) : nullptr;
        if (!kq || kq->in[0] != kv || kq->in[1] != qp->out[0]) return false;
        OpNode *gm = c.take(OpType::GET_MASK);
        if (!gm) return false;
        const auto &mp = gm->get_params<GetMaskParams>();
        if (mp.pos != *pos || gm->out[0]->m_shape != Shape{n_kv, bs, 1, 1}) return false;
        mask = mp.mask;
        OpNode *sm = c.take(OpType::SOFTMAX_EXT);
        if (!sm || sm->in[0] != kq->out[0] || sm->in[1] != gm->out[0] || sm->get_params<SoftmaxExtParams>().scale != want_scale || sm->get_params<SoftmaxExtParams>().max_bias != 0.0f) return false;
        TensorNode *vv = view_of(vc_dev, Shape{n_kv, hs, n_head_kv, 1}, Shape{es, es * n_ctx, es * n_ctx * hs, es * n_ctx * hs * n_head_kv}, 0);
        OpNode *pv = vv ? c.take(OpType::MAT_MUL) : nullptr;
        if (!pv || pv->in[0] != vv || pv->in[1] != sm->out[0]) return false;
        OpNode *mg = c.take(OpType::PERMUTE);
        if (!mg || mg->in[0] != pv->out[0] || mg->get_params<PermuteParams>().axes != heads_perm) return false;
        OpNode *ct = c.take(OpType::CONT);
        if (!ct || ct->in[0] != mg->out[0] || ct->out[0]->m_shape != Shape{hs * n_head, bs, 1, 1}) return false;
        TensorNode *o = mat_mul(W.wo, ct->out[0]);
        OpNode *res = c.take(OpType::ADD);
        if (!o || !res || res->in[0] != x || res->in[1] != o) return false;
        TensorNode *a = res->out[0];
        // ---- FFN::build
        TensorNode *n2 = rms(W.ffn_norm, a);
        if (!n2) return false;
        TensorNode *g = mat_mul(W.wg, n2), *u = g ? mat_mul(W.wu, n2) : nullptr;
        OpNode *sh = c.take(OpType::SILU_HADAMARD);
        if (!u || !sh || sh->in[0] != g || sh->in[1] != u) return false;
        TensorNode *d = mat_mul(W.wd, sh->out[0]);
        OpNode *res2 = c.take(OpType::ADD);
        if (!d || !res2 || res2->in[0] != a || res2->in[1] != d) return false;
        x = res2->out[0];
    }
    out.lm_head = false;
    out.logits  = nullptr;
    if (!c.done()) { // final norm + lm_head (tied: the token table)
        TensorNode *n = rms(T.output_norm, x);
        TensorNode *lg = n ? mat_mul(T.output ? T.output : T.token_embd, n) : nullptr;
        if (!lg || !c.done()) return false;
        out.lm_head = true;
        out.logits  = lg;
    }
    if (!pos) return false;
    for (size_t i = 1; i < bs; i++) if ((*pos)[i] != (*pos)[0] + (int)i) return false; // the KV append is one contiguous block
    // the lowered forward appends at slot pos[0] and HIPKV::advance then adds bs to the cache position: the two agree only when
    // the graph was built for the current position (a re-run of an earlier position goes op by op, which writes where it is told)
    if ((size_t)(*pos)[0] != m_kv->position()) return false;
    out.tokens.assign(tokens.begin(), tokens.end());
    out.pos.assign(pos->begin(), pos->end());
    out.tree.clear();
    if (mask && !mask->mask.empty()) {
        out.tree.resize(bs * bs);
        for (size_t i = 0; i < bs; i++) for (size_t j = 0; j < bs; j++) out.tree[i * bs + j] = mask->mask[i][j] ? 1 : 0;
    }
    return true;
}

void HIPBackend::plan(std::vector<std::shared_ptr<OpNode>> &ops) {
    // (the reference sizes its CPU work buffer here, src/backend/ggml/ggml.cpp:30-109; on the device that scratch lives in
    // the kernels' LDS)
    n_plans++;
    m_low.ok = m_fused && match_canonical(ops, m_low);
    if (m_low.ok) n_lowered++;
}

void HIPBackend::run_lowered() {
    POWERSERVE_ASSERT(m_low.ok);
    check(ps_hip_model_forward_lowered(m_model, m_low.tokens.data(), (int)m_low.tokens.size(), m_low.pos.data(),
                                       m_low.tree.empty() ? nullptr : m_low.tree.data(), m_low.lm_head ? 1 : 0), "lowered forward");
    if (m_low.logits) { // the graph's logits tensor is the model's buffer
        auto &t = *m_low.logits;
        Stride st; st[0] = sizeof(float); for (size_t i = 1; i < 4; i++) st[i] = st[i - 1] * t.m_shape[i - 1];
        t.m_data = std::make_shared<HIPBuffer>(st, (void *)ps_hip_model_logits(m_model));
    }
    m_low.ok = false;
}

} // namespace hip

void Platform::init_hip_backend(const std::shared_ptr<ModelConfig> &config, const HyperParams &hparams, int device) {
    hip_backends.insert({config->model_id, std::make_unique<hip::HIPBackend>(config->llm, hparams, device)});
}

// ---------------------------------------------------------------- executor (src/executor/executor.cpp:23-235)
void Executor::allocate_buffers() {
    auto &be = *m_platform.hip_backends[m_graph.m_model_id];
    auto cstride = [](const Tensor &t, size_t es) { Stride s; s[0] = es; for (size_t i = 1; i < 4; i++) s[i] = s[i - 1] * t.m_shape[i - 1]; return s; };
    size_t need = 0;
    for (auto &t : m_graph.tensors)
        if (!t->m_data && !t->is_view()) need += (t->n_elements() * (t->m_dtype == DataType::INT64 ? 8 : 4) + 255) / 256 * 256;
    be.arena_reset();
    be.arena_reserve(need + 4096); // one device allocation at most (grown geometrically), not one malloc per intermediate
    for (auto &t : m_graph.tensors) {
        if (t->m_data) continue;
        size_t es;
        switch (t->m_dtype) {
        case DataType::FP32: case DataType::INT32: es = 4; break;
        case DataType::INT64: es = 8; break;
        default: POWERSERVE_ABORT("could not allocate buffer for data type");
        }
        if (t->is_view()) {
            POWERSERVE_ASSERT(t->alias_of->m_data != nullptr, "a view was created before its source had storage");
            t->m_data = std::make_shared<HIPBuffer>(cstride(*t, es), t->alias_of->get<HIPBuffer>().m_data);
        } else {
            t->m_data = std::make_shared<HIPBuffer>(cstride(*t, es), be.arena_alloc(t->n_elements() * es));
        }
    }
}
void Executor::plan() { m_platform.hip_backends[m_graph.m_model_id]->plan(m_graph.ops); m_planned = true; }

void Executor::run() {
    auto &be = *m_platform.hip_backends[m_graph.m_model_id];
    if (!m_planned) plan(); // (the reference plans inside run(), executor.cpp:79; a caller may plan first to learn whether buffers are needed)
    m_planned = false;
    if (be.lowered()) { be.run_lowered(); return; }
    for (auto &op : m_graph.ops) {
        switch (op->op) {
        case OpType::GET_EMBEDDING: be.get_embedding(op->output(), op->in[0], op->get_params<GetEmbeddingParams>().tokens); break;
        case OpType::ADD: be.add(op->output(), op->in[0], op->in[1]); break;
        case OpType::MAT_MUL: be.matmul(op->output(), op->in[0], op->in[1]); break;
        case OpType::RMS_NORM: be.rmsnorm(op->output(), op->in[0], op->in[1], op->get_params<RMSNormParams>().eps); break;
        case OpType::SILU_HADAMARD: be.silu_hadamard(op->output(), op->in[0], op->in[1]); break;
        case OpType::ROPE: { auto &p = op->get_params<RopeParams>(); be.rope(op->out[0], op->in[0], p.pos, p.rope_cfg); } break;
        case OpType::SOFTMAX: be.softmax(op->output(), op->in[0]); break;
        case OpType::COPY: be.copy(op->in[0], op->in[1]); break;
        case OpType::PRINT: be.print(op->in[0], op->get_params<PrintParams>().size); break;
        case OpType::PERMUTE: be.permute(op->output(), op->in[0], op->get_params<PermuteParams>().axes); break;
        case OpType::CONT: be.cont(op->output(), op->in[0]); break;
        case OpType::VIEW: { // executed in the executor, like the reference (executor.cpp:194-199)
            auto out = op->output(); auto &p = op->get_params<ViewParams>();
            out->get<HIPBuffer>().m_stride = p.stride;
            out->get<HIPBuffer>().m_data   = (char *)out->get<HIPBuffer>().m_data + p.offset;
        } break;
        case OpType::SOFTMAX_EXT: { auto &p = op->get_params<SoftmaxExtParams>(); be.softmax_ext(op->output(), op->in[0], op->in[1], p.scale, p.max_bias); } break;
        case OpType::GET_MASK: { auto &p = op->get_params<GetMaskParams>(); be.get_mask(op->output(), p.pos, *p.mask); } break;
        case OpType::TRANSPOSE: be.transpose(op->output(), op->in[0]); break;
        default: POWERSERVE_ABORT("Unknown OpType: " + std::to_string((int)op->op));
        }
    }
}

} // namespace powerserve
