#include <cstdarg>
#include <cstdio>
// C-ABI (include/ps_hip.h): context, memory, weights and the reference-shaped operator entry points.
#include "ps_internal.h"
#include "ps_ops.h"
#include "ps_dev.h" // PS_CONTRACT_ON

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>

// ------------------------------------------------------------------ host restatement of the RoPE cache
// ggml_rope_cache_init (libs/ggml/src/ggml.c:15344-15358) + rope_yarn (:15319-15336) + corr dims
// (:15360-15366).  Computed on the host with the same libm calls and the same theta *= theta_scale
// recurrence as the reference, then uploaded; freq_factors is always NULL in PowerServe
// (backend/ggml/ggml_wrapper.cpp:104-106).
static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float)M_PI)) / (2 * logf(base));
}
void ps_rope_table_host(const ps_rope_params *rp, int64_t ne0, const int32_t *pos, int n_pos, float *table) {
    const float theta_scale = powf(rp->freq_base, -2.0f / rp->n_dims);
    float corr0 = fmaxf(0.f, floorf(rope_corr_dim(rp->n_dims, rp->n_ctx_orig, rp->beta_fast, rp->freq_base)));
    float corr1 = fminf((float)rp->n_dims - 1, ceilf(rope_corr_dim(rp->n_dims, rp->n_ctx_orig, rp->beta_slow, rp->freq_base)));
    for (int i = 0; i < n_pos; i++) {
        float *cache = table + (int64_t)i * ne0;
        volatile float theta = (float)(pos ? pos[i] : i);
        for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
            const float theta_extrap = theta;
            volatile float theta_interp = rp->freq_scale * theta_extrap;
            float th = theta_interp, mscale = rp->attn_factor;
            if (rp->ext_factor != 0.0f) {
                const float y = ((int)i0 / 2 - corr0) / fmaxf(0.001f, corr1 - corr0);
                const float ramp_mix = (1 - fminf(1, fmaxf(0, y))) * rp->ext_factor;
                volatile float t1 = theta_interp * (1 - ramp_mix);
                volatile float t2 = theta_extrap * ramp_mix;
                th = t1 + t2;
                mscale *= 1.0f + 0.1f * logf(1.0f / rp->freq_scale);
            }
            volatile float c = cosf(th), s = sinf(th);
            cache[i0 + 0] = c * mscale;
            cache[i0 + 1] = s * mscale;
            theta = theta * theta_scale;
        }
    }
}

static inline bool is_quant(int t) { return t == PS_Q4_0 || t == PS_Q8_0 || t == PS_Q4_K || t == PS_Q5_K || t == PS_Q6_K; }
static inline bool is_row_wave(int t) { return t == PS_Q5_K || t == PS_Q6_K; } // one wave per weight row (k_gemv6.hip)

// ------------------------------------------------------------------ device allocations (ps_internal.h)
namespace {
struct GuardRec { void *va; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; };
std::mutex g_guard_mu;
std::unordered_map<void *, GuardRec> g_guard;
int guard_align() { static const int a = getenv("PS_HIP_GUARD") ? atoi(getenv("PS_HIP_GUARD")) : 0; return a; } // 0: off; 1: on, hipMalloc's 256-byte alignment kept; 16 / 64 / ...: that alignment
bool guard_on() { return guard_align() != 0; }
} // namespace
hipError_t ps_dev_malloc(void **p, size_t bytes) {
    if (!guard_on()) return hipMalloc(p, bytes);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    const size_t al = guard_align() >= 16 ? (size_t)guard_align() : 256;
    const size_t user = (bytes + al - 1) / al * al, mapped = (user + gran - 1) / gran * gran, reserved = mapped + gran; // (the last granule stays unmapped)
    GuardRec r{nullptr, reserved, mapped, {}};
    if ((e = hipMemAddressReserve(&r.va, reserved, gran, nullptr, 0)) != hipSuccess) return e;
    if ((e = hipMemCreate(&r.h, mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(r.va, reserved); return e; }
    if ((e = hipMemMap(r.va, mapped, 0, r.h, 0)) != hipSuccess) { (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, reserved); return e; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(r.va, mapped, &acc, 1)) != hipSuccess) { (void)hipMemUnmap(r.va, mapped); (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, reserved); return e; }
    *p = (char *)r.va + (mapped - user);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[*p] = r;
    return hipSuccess;
}
hipError_t ps_dev_free(void *p) {
    if (!p) return hipSuccess;
    if (!guard_on()) return hipFree(p);
    GuardRec r;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guard.find(p);
        if (it == g_guard.end()) return hipFree(p);
        r = it->second;
        g_guard.erase(it);
    }
    // the range is never handed back: with hipMemUnmap / hipMemRelease / hipMemAddressFree behind every free, later allocations of the same process read
    // wrong data now and then (ops tests failed at random under recycling and pass without it); a debugging run can afford the memory
    return hipSuccess;
}


extern "C" {

int ps_hip_abi_version(void) { return PS_HIP_ABI_VERSION; }
int ps_hip_build_contract(void) { return PS_CONTRACT_ON ? 1 : 0; }

int ps_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


int ps_hip_create(int device, ps_hip_ctx **out) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) return 1;
    if (hipSetDevice(device) != hipSuccess) return 1;
    auto c    = new ps_hip_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return 1; }
    *out = c;
    return 0;
}

void ps_hip_destroy(ps_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->act_buf) (void)ps_dev_free(c->act_buf);
    if (c->i32_buf) (void)ps_dev_free(c->i32_buf);
    if (c->u8_buf) (void)ps_dev_free(c->u8_buf);
    if (c->rope_buf) (void)ps_dev_free(c->rope_buf);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *ps_hip_last_error(const ps_hip_ctx *c) { return c ? c->err.c_str() : "no context"; }

int ps_hip_device_name(const ps_hip_ctx *c, char *buf, size_t cap) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return 1;
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int ps_hip_malloc(ps_hip_ctx *c, size_t bytes, void **dptr) {
    PS_CHECK(c, hipSetDevice(c->device));
    PS_CHECK(c, ps_dev_malloc(dptr, bytes ? bytes : 16));
    return 0;
}
int ps_hip_free(ps_hip_ctx *c, void *dptr) {
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    PS_CHECK(c, ps_dev_free(dptr));
    return 0;
}
int ps_hip_memcpy_h2d(ps_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    PS_CHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int ps_hip_memcpy_d2h(ps_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    PS_CHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int ps_hip_memset(ps_hip_ctx *c, void *dst, int value, size_t bytes) {
    PS_CHECK(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return 0;
}
int ps_hip_sync(ps_hip_ctx *c) {
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    PS_CHECK(c, hipGetLastError());
    return 0;
}
void *ps_hip_stream(ps_hip_ctx *c) { return (void *)c->stream; }

int ps_hip_event_create(ps_hip_ctx *c, void **ev) {
    hipEvent_t e;
    PS_CHECK(c, hipEventCreate(&e));
    *ev = (void *)e;
    return 0;
}
int ps_hip_event_record(ps_hip_ctx *c, void *ev) {
    PS_CHECK(c, hipEventRecord((hipEvent_t)ev, c->stream));
    return 0;
}
int ps_hip_event_elapsed_ms(ps_hip_ctx *c, void *a, void *b, float *ms) {
    PS_CHECK(c, hipEventSynchronize((hipEvent_t)b));
    PS_CHECK(c, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return 0;
}
int ps_hip_event_destroy(ps_hip_ctx *c, void *ev) {
    PS_CHECK(c, hipEventDestroy((hipEvent_t)ev));
    return 0;
}

// ------------------------------------------------------------------ type helpers (ggml.c:681-1015)
size_t ps_hip_row_size(int t, int64_t k) {
    switch (t) {
    case PS_F32: case PS_I32: return (size_t)k * 4;
    case PS_F16: return (size_t)k * 2;
    case PS_Q4_0: return (size_t)(k / 32) * 18;
    case PS_Q8_0: return (size_t)(k / 32) * 34;
    case PS_Q4_K: return (size_t)(k / 256) * 144;
    case PS_Q5_K: return (size_t)(k / 256) * 176;
    case PS_Q6_K: return (size_t)(k / 256) * 210;
    case PS_Q8_K: return (size_t)(k / 256) * 292;
    }
    return 0;
}
int ps_hip_vec_dot_type(int t) {
    switch (t) {
    case PS_Q4_0: case PS_Q8_0: return PS_Q8_0;
    case PS_Q4_K: case PS_Q5_K: case PS_Q6_K: return PS_Q8_K;
    }
    return t;
}

// ------------------------------------------------------------------ weights
int ps_hip_weight_upload(ps_hip_ctx *c, int dtype, const void *host, int64_t K, int64_t N, ps_weight **out) {
    *out = nullptr;
    if (!(is_quant(dtype) || dtype == PS_F32)) PS_FAIL(c, "weight_upload: unsupported dtype");
    const int64_t blk = (dtype == PS_Q4_K || dtype == PS_Q5_K || dtype == PS_Q6_K) ? 256 : (dtype == PS_F32 ? 1 : 32);
    if (K % blk) PS_FAIL(c, "weight_upload: K is not a multiple of the block size");
    if (PS_CONTRACT_ON && dtype == PS_Q5_K) PS_FAIL(c, "weight_upload: this is the PS_CONTRACT build (the reference's stock -ffp-contract=fast build); Q5_K's fused summs is not implemented in it");
    PS_CHECK(c, hipSetDevice(c->device));
    auto w        = new ps_weight();
    w->dtype      = dtype;
    w->K          = K;
    w->N          = N;
    w->gguf_bytes = (uint64_t)N * ps_hip_row_size(dtype, K);
    const size_t raw = (size_t)w->gguf_bytes;
    auto fail = [&](const char *m) { ps_hip_weight_free(c, w); c->err = m; return 1; };
    if (dtype == PS_F32) {
        if (ps_dev_malloc((void **)&w->qs, raw) != hipSuccess) return fail("weight_upload: hipMalloc");
        if (hipMemcpyAsync(w->qs, host, raw, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail("weight_upload: copy");
        PS_CHECK(c, hipStreamSynchronize(c->stream));
        *out = w;
        return 0;
    }
    uint8_t *tmp = nullptr;
    if (ps_dev_malloc((void **)&tmp, raw + 64) != hipSuccess) return fail("weight_upload: ps_dev_malloc(tmp)");
    size_t qs_b = 0, aux_b = 0, qh_b = 0, sc_b = 0;
    if (dtype == PS_Q4_0 || dtype == PS_Q8_0 || dtype == PS_Q4_K) { // lane-major repack: 1 KiB units per row group
        const int64_t rg = ps_w_rg(dtype), ng = (N + rg - 1) / rg, nu = (K + ps_w_unit(dtype) - 1) / ps_w_unit(dtype);
        qs_b  = (size_t)(ng * nu) * 1024;
        aux_b = (size_t)(ng * nu) * (size_t)rg * (dtype == PS_Q4_K ? 16 : 8);
    }
    if (dtype == PS_Q6_K) { qs_b = (size_t)N * K / 2; qh_b = (size_t)N * K / 4; sc_b = (size_t)N * K / 16; aux_b = (size_t)N * (K / 256) * 2; }
    if (dtype == PS_Q5_K) { qs_b = (size_t)N * K / 2; qh_b = (size_t)N * K / 8; sc_b = (size_t)N * (K / 256) * 16; aux_b = 0; }
    bool ok = ps_dev_malloc((void **)&w->qs, qs_b + 64) == hipSuccess && ps_dev_malloc((void **)&w->aux, aux_b + 64) == hipSuccess;
    if (ok && qh_b) ok = ps_dev_malloc((void **)&w->qh, qh_b + 64) == hipSuccess && ps_dev_malloc((void **)&w->sc, sc_b + 64) == hipSuccess;
    if (!ok) { (void)ps_dev_free(tmp); return fail("weight_upload: ps_dev_malloc(planes)"); }
    if (hipMemcpyAsync(tmp, host, raw, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)ps_dev_free(tmp); return fail("weight_upload: copy"); }
    psk_repack_weight(c->stream, dtype, tmp, K, N, w);
    hipError_t e = hipStreamSynchronize(c->stream);
    (void)ps_dev_free(tmp);
    if (e != hipSuccess) return fail("weight_upload: repack");
    *out = w;
    return 0;
}

void ps_hip_weight_free(ps_hip_ctx *c, ps_weight *w) {
    if (!w) return;
    (void)c;
    if (w->qs) (void)ps_dev_free(w->qs);
    if (w->aux) (void)ps_dev_free(w->aux);
    if (w->qh) (void)ps_dev_free(w->qh);
    if (w->sc) (void)ps_dev_free(w->sc);
    delete w;
}
uint64_t ps_hip_weight_gguf_bytes(const ps_weight *w) { return w->gguf_bytes; }
int ps_hip_weight_dtype(const ps_weight *w) { return w->dtype; }

// ------------------------------------------------------------------ scratch helpers
static int ensure(ps_hip_ctx *c, void **buf, size_t *cap, size_t need) {
    if (*cap >= need) return 0;
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    if (*buf) PS_CHECK(c, ps_dev_free(*buf));
    *buf = nullptr;
    *cap = 0;
    PS_CHECK(c, ps_dev_malloc(buf, need));
    *cap = need;
    return 0;
}
static int stage_i32(ps_hip_ctx *c, const int32_t *host, int n, int32_t **dev) {
    if (ensure(c, (void **)&c->i32_buf, &c->i32_cap, (size_t)n * 4 + 16)) return 1;
    PS_CHECK(c, hipMemcpyAsync(c->i32_buf, host, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    PS_CHECK(c, hipStreamSynchronize(c->stream)); // host buffer may be a temporary
    *dev = c->i32_buf;
    return 0;
}

// ------------------------------------------------------------------ ops
int ps_hip_quantize_act(ps_hip_ctx *c, int vdt, const float *x, int64_t K, int64_t rows, void *out_blocks) {
    if (vdt != PS_Q8_0 && vdt != PS_Q8_K) PS_FAIL(c, "quantize_act: vdt must be Q8_0 or Q8_K");
    if (K % (vdt == PS_Q8_0 ? 32 : 256)) PS_FAIL(c, "quantize_act: K not a multiple of the block size");
    if (ensure(c, &c->act_buf, &c->act_cap, ps_act_bytes(K, rows))) return 1;
    ps_act a = ps_act_carve(c->act_buf, K, rows);
    psk_quantize_act(c->stream, vdt, 0, x, nullptr, nullptr, 0.f, K, rows, a);
    psk_pack_act_blocks(c->stream, vdt, a, K, rows, out_blocks);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_mul_mat(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *src0, const ps_tensor *src1) {
    // checks mirror powerserve_compute_forward_mul_mat's asserts (ggml.c:13457-13470) and Graph::mat_mul
    if (src0->ne[0] != src1->ne[0]) PS_FAIL(c, "mul_mat: ne00 != ne10");
    if (dst->ne[0] != src0->ne[1] || dst->ne[1] != src1->ne[1] || dst->ne[2] != src1->ne[2] || dst->ne[3] != src1->ne[3])
        PS_FAIL(c, "mul_mat: bad dst shape");
    if (src1->dtype != PS_F32 || dst->dtype != PS_F32) PS_FAIL(c, "mul_mat: src1/dst must be F32");
    if (src1->nb[0] != 4 || dst->nb[0] != 4) PS_FAIL(c, "mul_mat: permuted src1/dst not supported");
    if (is_quant(src0->dtype)) {
        const ps_weight *w = (const ps_weight *)src0->data;
        if (!w || w->dtype != src0->dtype || w->K != src0->ne[0] || w->N != src0->ne[1]) PS_FAIL(c, "mul_mat: weight handle mismatch");
        const int64_t K = w->K, bs = src1->ne[1] * src1->ne[2] * src1->ne[3];
        if (src1->nb[1] != (uint64_t)K * 4 || dst->nb[1] != (uint64_t)w->N * 4) PS_FAIL(c, "mul_mat: quantized path needs contiguous rows");
        const int vdt = ps_hip_vec_dot_type(w->dtype);
        if (ensure(c, &c->act_buf, &c->act_cap, ps_act_bytes(K, bs))) return 1;
        ps_act a = ps_act_carve(c->act_buf, K, bs);
        if (is_row_wave(w->dtype) && bs == 1) { // single column: producer / chain-wave kernel, quantizer in its prologue (k_gemvk.hip)
            psk_gemv_args g{};
            g.n_w = 1; g.w[0] = w; g.out[0] = (float *)dst->data; g.ldo[0] = w->N;
            g.pro = 2; g.pro_x = (const float *)src1->data;
            const int rc = psk_gemvk(c->stream, c->n_cu, g, a, K);
            if (rc == 0) { PS_CHECK(c, hipGetLastError()); return 0; }
            if (rc != -1) { c->err = "mul_mat: Q5_K / Q6_K mat-vec launch rc=" + std::to_string(rc); return 2; }
        }
        if (is_row_wave(w->dtype)) { // one wave per row (k_gemv6.hip)
            psk_quantize_act(c->stream, vdt, 0, (const float *)src1->data, nullptr, nullptr, 0.f, K, bs, a);
            psk_gemv6_args g6{w, (float *)dst->data, w->N, nullptr, nullptr};
            if (int rc = psk_gemv6(c->stream, c->n_cu, g6, a, K, bs)) { c->err = "mul_mat: Q5_K / Q6_K launch rc=" + std::to_string(rc); return 2; }
            PS_CHECK(c, hipGetLastError());
            return 0;
        }
        if (bs >= 2) { // batches: quantize once, 8 / 16 columns per workgroup (psk_gemm8)
            psk_quantize_act(c->stream, vdt, 0, (const float *)src1->data, nullptr, nullptr, 0.f, K, bs, a);
            psk_gemv_args g{};
            g.n_w = 1; g.w[0] = w; g.out[0] = (float *)dst->data; g.ldo[0] = w->N;
            const int rc = psk_gemm8(c->stream, c->n_cu, g, a, K, bs);
            if (rc == 0) { PS_CHECK(c, hipGetLastError()); return 0; }
            if (rc != -1) { c->err = "mul_mat: gemm launch rc=" + std::to_string(rc); return 2; }
        }
        for (int64_t c0 = 0; c0 < bs; c0 += 4) { // <= 4 columns per GEMV launch; activation quantized in its prologue
            const int64_t nb = bs - c0 < 4 ? bs - c0 : 4;
            psk_gemv_args g{};
            g.n_w = 1; g.w[0] = w; g.out[0] = (float *)dst->data + c0 * w->N; g.ldo[0] = w->N;
            g.pro = 2; g.pro_x = (const float *)src1->data + c0 * K;
            if (int rc = psk_gemv(c->stream, c->n_cu, g, a, vdt, K, nb)) { c->err = "mul_mat: gemv launch rc=" + std::to_string(rc); return 2; }
        }
    } else if (src0->dtype == PS_F32) {
        if (src0->nb[0] != 4) PS_FAIL(c, "mul_mat: transposed src0 not supported");
        if (src1->ne[2] % src0->ne[2] || src1->ne[3] % src0->ne[3]) PS_FAIL(c, "mul_mat: src0 not broadcastable");
        psl_mul_mat_f32(c->stream, dst, src0, src1);
    } else {
        PS_FAIL(c, "mul_mat: unsupported src0 dtype");
    }
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_rms_norm(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *src, const ps_tensor *weight, float eps) {
    if (src->dtype != PS_F32 || src->nb[0] != 4 || dst->nb[0] != 4) PS_FAIL(c, "rms_norm: F32 rows required");
    if (!(eps > 0.0f)) PS_FAIL(c, "rms_norm: eps must be > 0");
    if (weight && weight->ne[0] != src->ne[0]) PS_FAIL(c, "rms_norm: weight length mismatch");
    psl_rms_norm(c->stream, dst, src, weight ? (const float *)weight->data : nullptr, eps);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_rope(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *src, const int32_t *pos, int n_pos, const ps_rope_params *rp) {
    if (src->ne[2] != n_pos) PS_FAIL(c, "rope: ne2 != number of positions");
    if (rp->n_dims > src->ne[0] || rp->n_dims % 2) PS_FAIL(c, "rope: bad n_dims");
    const size_t bytes = (size_t)n_pos * src->ne[0] * 4;
    std::vector<float> tab((size_t)n_pos * src->ne[0]);
    ps_rope_table_host(rp, src->ne[0], pos, n_pos, tab.data());
    if (ensure(c, (void **)&c->rope_buf, &c->rope_cap, bytes)) return 1;
    PS_CHECK(c, hipMemcpyAsync(c->rope_buf, tab.data(), bytes, hipMemcpyHostToDevice, c->stream));
    PS_CHECK(c, hipStreamSynchronize(c->stream));
    psl_rope(c->stream, dst, src, c->rope_buf, rp->n_dims, (rp->mode & 2) ? 1 : 0);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_soft_max(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *src) { return ps_hip_softmax_ext(c, dst, src, nullptr, 1.0f, 0.0f); } // ggml.c:15071-15076

int ps_hip_softmax_ext(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *src, const ps_tensor *mask, float scale, float max_bias) {
    if (max_bias != 0.0f) PS_FAIL(c, "softmax_ext: ALiBi (max_bias != 0) is not on PowerServe's path");
    if (src->ne[0] * 4 > 150 * 1024) PS_FAIL(c, "softmax_ext: row too long for the LDS-resident kernel");
    psl_softmax_ext(c->stream, dst, src, mask ? (const float *)mask->data : nullptr, scale);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_add(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *a, const ps_tensor *b) {
    for (int i = 0; i < 4; i++)
        if (a->ne[i] % b->ne[i]) PS_FAIL(c, "add: b is not repeat-broadcastable into a");
    psl_add(c->stream, dst, a, b);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_dup(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *src) {
    const int64_t nd = dst->ne[0] * dst->ne[1] * dst->ne[2] * dst->ne[3], ns = src->ne[0] * src->ne[1] * src->ne[2] * src->ne[3];
    if (nd != ns) PS_FAIL(c, "dup: element counts differ");
    if (dst->dtype != PS_F32 || src->dtype != PS_F32) PS_FAIL(c, "dup: F32 only");
    psl_dup(c->stream, dst, src);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_silu_hadamard(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *gate, const ps_tensor *up) {
    const int64_t n = gate->ne[0] * gate->ne[1] * gate->ne[2] * gate->ne[3];
    psl_silu_hadamard(c->stream, (float *)dst->data, (const float *)gate->data, (const float *)up->data, n);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_get_embedding(ps_hip_ctx *c, const ps_tensor *dst, const ps_tensor *weight, const int32_t *tokens, int n) {
    const ps_weight *w = (const ps_weight *)weight->data;
    if (!w || dst->ne[0] != w->K || dst->ne[1] != n) PS_FAIL(c, "get_embedding: shape mismatch");
    for (int i = 0; i < n; i++)
        if (tokens[i] < 0 || tokens[i] >= w->N) PS_FAIL(c, "get_embedding: token id out of range");
    int32_t *td;
    if (stage_i32(c, tokens, n, &td)) return 1;
    psl_get_rows(c->stream, w, td, n, (float *)dst->data);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_get_mask(ps_hip_ctx *c, const ps_tensor *dst, const int32_t *pos, int n_pos, const uint8_t *tree) {
    if (dst->ne[1] != n_pos) PS_FAIL(c, "get_mask: ne1 != batch");
    int32_t *pd;
    if (stage_i32(c, pos, n_pos, &pd)) return 1;
    const uint8_t *td = nullptr;
    if (tree) {
        if (ensure(c, (void **)&c->u8_buf, &c->u8_cap, (size_t)n_pos * n_pos)) return 1;
        PS_CHECK(c, hipMemcpyAsync(c->u8_buf, tree, (size_t)n_pos * n_pos, hipMemcpyHostToDevice, c->stream));
        PS_CHECK(c, hipStreamSynchronize(c->stream));
        td = c->u8_buf;
    }
    psl_get_mask(c->stream, (float *)dst->data, dst->ne[0], n_pos, pd, td);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

int ps_hip_argmax(ps_hip_ctx *c, const float *src, int64_t n, int64_t rows, int32_t *out_dev) {
    psl_argmax(c->stream, src, n, rows, out_dev);
    PS_CHECK(c, hipGetLastError());
    return 0;
}

} // extern "C"
static char g_last_kernel[128] = "";
void psk_note_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}
const char *psk_last_kernel() { return g_last_kernel; }
extern "C" {
const char *ps_hip_last_matmul_kernel(void) { return psk_last_kernel(); }
int ps_hip_debug_timeline(ps_hip_ctx *ctx, int key, uint64_t *host_out, int n_words) {
    if (!ctx) return 1;
    if (!PS_TIMELINE && key >= 0) PS_FAIL(ctx, "debug_timeline: this library is built without the in-kernel timeline marks (they cost the decode path 3 %); "
                                               "use lib/libps_hip_timeline.so (python -m powerserve_amd.build --timeline; PS_HIP_LIB)");
    if (host_out) PS_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return psk_gemv_debug(key, host_out, n_words);
}

int ps_hip_debug_set(int key, int value) {
    extern int g_g4_cfg, g_g4_flags, g_g4k_par, g_f16_variant, g_force_attn_timeout, g_g4k_cbx, g_qa_force, g_kv_stream_force;
    if (key == 1) { g_g4_cfg = value; return 0; }
    if (key == 2) { g_g4_flags = value; return 0; }
    if (key == 3) { g_g4k_par = value; return 0; }
    if (key == 4) { g_f16_variant = value; return 0; }
    if (key == 5) { g_force_attn_timeout = value; return 0; }
    if (key == 6) { g_g4k_cbx = value; return 0; }
    if (key == 7) { g_qa_force = value; return 0; }
    if (key == 9) { g_kv_stream_force = value; return 0; } // single-token attention: the cached K / V with plain (0) / non-temporal (1) loads whatever the cache's size; -1: by size (model.hip) // the fused QKV + attention launch wherever it is covered (default: only where its grid fills 3/4 of the chip)
    return 1;
}

} // extern "C"
