// Reference-shaped operator kernels (strided tensors, F32): the op-by-op path HIPBackend takes for any
// graph the fused planner does not recognise, and the parity anchor for each fused kernel.
// Built with -ffp-contract=off; every op rounds where the reference's C source rounds.
#include "ps_dev.h"
#include "ps_expf.h"
#include "ps_internal.h"

struct TDesc { // device view of a ps_tensor
    char *data;
    int64_t ne[4];
    int64_t nb[4];
};

static inline TDesc tdesc(const ps_tensor *t) {
    TDesc d;
    d.data = (char *)t->data;
    for (int i = 0; i < 4; i++) { d.ne[i] = t->ne[i]; d.nb[i] = (int64_t)t->nb[i]; }
    return d;
}

namespace {

// ---------------------------------------------------------------- F32 x F32 mat-mul with ggml broadcast
// ggml_vec_dot_f32 (ggml.c:2092-2133) under powerserve_compute_forward_mul_mat with F32 traits: used for K·q and
// V·softmax (model/module/norm_attention.cpp:115-147).  A half-wave (32 lanes) per output element owns the 32
// fp32 chains of the reference's AVX build (elements 32*i + c), reduces them in GGML_F32x8_REDUCE order and adds
// the n % 32 leftovers one by one: bit-exact.
__global__ __launch_bounds__(256) void mul_mat_f32_kernel(TDesc dst, TDesc a, TDesc b) {
    const int c = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int64_t n_out = dst.ne[0] * dst.ne[1] * dst.ne[2] * dst.ne[3];
    const int64_t r2 = b.ne[2] / a.ne[2], r3 = b.ne[3] / a.ne[3];
    const int64_t n = a.ne[0], np = n & ~(int64_t)31;
    for (int64_t o0 = (int64_t)blockIdx.x * 8; o0 < n_out; o0 += (int64_t)gridDim.x * 8) {
        const int64_t o = o0 + hw;
        const bool live = o < n_out;
        const int64_t oo = live ? o : 0;
        int64_t i0 = oo % dst.ne[0], rest = oo / dst.ne[0];
        const int64_t i1 = rest % dst.ne[1]; rest /= dst.ne[1];
        const int64_t i2 = rest % dst.ne[2], i3 = rest / dst.ne[2];
        const float *x = (const float *)(a.data + i0 * a.nb[1] + (i2 / r2) * a.nb[2] + (i3 / r3) * a.nb[3]);
        const float *y = (const float *)(b.data + i1 * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3]);
        float s = 0.f;
        for (int64_t k = c; k < np; k += 32) s = __fmaf_rn(x[k], y[k], s);
        s = reduce_f32x8x4(s);
        for (int64_t k = np; k < n; k++) s = ps_dot_left(s, x[k], y[k], (int)(k - np), (int)(n - np));
        if (live && c == 0) *(float *)(dst.data + i0 * dst.nb[0] + i1 * dst.nb[1] + i2 * dst.nb[2] + i3 * dst.nb[3]) = s;
    }
}

// ---------------------------------------------------------------- RMSNorm (ggml.c:12667-12720)
__global__ __launch_bounds__(256) void rms_norm_kernel(TDesc dst, TDesc src, const float *w, float eps) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.x; // over ne1*ne2*ne3
    const int64_t i1 = row % src.ne[1], i2 = (row / src.ne[1]) % src.ne[2], i3 = row / (src.ne[1] * src.ne[2]);
    const float *x = (const float *)(src.data + i1 * src.nb[1] + i2 * src.nb[2] + i3 * src.nb[3]);
    float *y       = (float *)(dst.data + i1 * dst.nb[1] + i2 * dst.nb[2] + i3 * dst.nb[3]);
    const int64_t n = src.ne[0];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)__fmul_rn(x[i], x[i]);
    s = wave_sum_d(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const double tot  = (red[0] + red[1]) + (red[2] + red[3]);
    const float mean  = (float)(tot / (double)n);
    const float scale = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, eps)));
    for (int64_t i = threadIdx.x; i < n; i += 256) y[i] = w ? __fmul_rn(x[i], __fmul_rn(w[i], scale)) : __fmul_rn(x[i], scale);
}

// ---------------------------------------------------------------- RoPE (ggml.c:15368-15491)
// cache: [npos][ne0] (cos, sin) pairs built on the host with the reference's recurrence.
__global__ void rope_kernel(TDesc dst, TDesc src, const float *cache, int n_dims, int neox) {
    const int64_t half = n_dims / 2;
    const int64_t per_row = src.ne[0] / 2; // pairs incl. pass-through tail
    const int64_t total = per_row * src.ne[1] * src.ne[2] * src.ne[3];
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pi = o % per_row; int64_t rest = o / per_row;
        const int64_t i1 = rest % src.ne[1]; rest /= src.ne[1];
        const int64_t i2 = rest % src.ne[2], i3 = rest / src.ne[2];
        const char *sb = src.data + i1 * src.nb[1] + i2 * src.nb[2] + i3 * src.nb[3];
        char *db       = dst.data + i1 * dst.nb[1] + i2 * dst.nb[2] + i3 * dst.nb[3];
        const int64_t i0 = 2 * pi;
        if (i0 >= n_dims) { // elements beyond n_dims are copied (ggml.c:15480-15487)
            *(float *)(db + i0 * 4) = *(const float *)(sb + i0 * 4);
            *(float *)(db + i0 * 4 + 4) = *(const float *)(sb + i0 * 4 + 4);
            continue;
        }
        const float c = cache[i2 * src.ne[0] + i0], s = cache[i2 * src.ne[0] + i0 + 1];
        const int64_t ia = neox ? pi : i0, ib = neox ? pi + half : i0 + 1;
        const float x0 = *(const float *)(sb + ia * 4), x1 = *(const float *)(sb + ib * 4);
        float ra, rb;
        ps_rope_pair(x0, x1, c, s, ra, rb);
        *(float *)(db + ia * 4) = ra;
        *(float *)(db + ib * 4) = rb;
    }
}

// ---------------------------------------------------------------- softmax_ext (ggml.c:14846-14940), max_bias == 0
// one workgroup per row, row held in LDS.  Exactly ggml_vec_soft_max_f32 (ggml.c:2814-2863): ggml_v_expf on the
// groups of 8 with the in-group sum tree, libm expf on the n % 8 tail, double row sum, p = e * (float)(1/sum).
__global__ __launch_bounds__(256) void softmax_ext_kernel(TDesc dst, TDesc src, const float *mask, float scale) {
    extern __shared__ float wp[];
    __shared__ float redf[4];
    __shared__ double redd[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.x, nc = src.ne[0], n8 = nc & ~(int64_t)7;
    const float *sp = (const float *)(src.data + row * src.nb[1]);
    float *dp       = (float *)(dst.data + row * dst.nb[1]);
    const float *mp = mask ? mask + (row % src.ne[1]) * nc : nullptr;
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < nc; i += 256) {
        float v = __fmul_rn(sp[i], scale);
        if (mp) v = __fadd_rn(v, mp[i]); // slope == 1
        wp[i] = v;
        mx    = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double sum = 0.0;
    for (int64_t g = threadIdx.x; g * 8 < n8; g += 256) {
        float v[8];
#pragma unroll
        for (int l = 0; l < 8; l++) v[l] = wp[g * 8 + l];
        ps_v_expf_n<8>(v, mx);
#pragma unroll
        for (int l = 0; l < 8; l++) wp[g * 8 + l] = v[l];
        const float a0 = __fadd_rn(v[4], v[0]), a1 = __fadd_rn(v[5], v[1]), a2 = __fadd_rn(v[6], v[2]), a3 = __fadd_rn(v[7], v[3]);
        sum += (double)__fadd_rn(__fadd_rn(a0, a2), __fadd_rn(a1, a3));
    }
    if (threadIdx.x == 0)
        for (int64_t i = n8; i < nc; i++) { const float e = ps_expf_glibc(__fsub_rn(wp[i], mx)); wp[i] = e; sum += (double)e; }
    sum = wave_sum_d(sum);
    if (lane == 0) redd[wave] = sum;
    __syncthreads();
    const double tot = (redd[0] + redd[1]) + (redd[2] + redd[3]);
    const float inv  = (float)(1.0 / tot);
    for (int64_t i = threadIdx.x; i < nc; i += 256) dp[i] = __fmul_rn(wp[i], inv);
}

// ---------------------------------------------------------------- add with ggml repeat-broadcast of b (ggml.c:10042-10115)
__global__ void add_kernel(TDesc dst, TDesc a, TDesc b) {
    const int64_t total = dst.ne[0] * dst.ne[1] * dst.ne[2] * dst.ne[3];
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = o % dst.ne[0]; int64_t rest = o / dst.ne[0];
        const int64_t i1 = rest % dst.ne[1]; rest /= dst.ne[1];
        const int64_t i2 = rest % dst.ne[2], i3 = rest / dst.ne[2];
        const float av = *(const float *)(a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
        const float bv = *(const float *)(b.data + (i0 % b.ne[0]) * b.nb[0] + (i1 % b.ne[1]) * b.nb[1] +
                                          (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3]);
        *(float *)(dst.data + i0 * dst.nb[0] + i1 * dst.nb[1] + i2 * dst.nb[2] + i3 * dst.nb[3]) = __fadd_rn(av, bv);
    }
}

// ---------------------------------------------------------------- dup: same-type strided copy (ggml.c:9519-9555).
// Like ggml's dup, src and dst only need the same number of elements; both are walked in their own
// logical (ne) order.
__global__ void dup_kernel(TDesc dst, TDesc src) {
    const int64_t total = src.ne[0] * src.ne[1] * src.ne[2] * src.ne[3];
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = o;
        const int64_t s0 = r % src.ne[0]; r /= src.ne[0];
        const int64_t s1 = r % src.ne[1]; r /= src.ne[1];
        const int64_t s2 = r % src.ne[2], s3 = r / src.ne[2];
        r = o;
        const int64_t d0 = r % dst.ne[0]; r /= dst.ne[0];
        const int64_t d1 = r % dst.ne[1]; r /= dst.ne[1];
        const int64_t d2 = r % dst.ne[2], d3 = r / dst.ne[2];
        *(float *)(dst.data + d0 * dst.nb[0] + d1 * dst.nb[1] + d2 * dst.nb[2] + d3 * dst.nb[3]) =
            *(const float *)(src.data + s0 * src.nb[0] + s1 * src.nb[1] + s2 * src.nb[2] + s3 * src.nb[3]);
    }
}

__global__ void silu_hadamard_kernel(float *out, const float *g, const float *u, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float val = g[i];
        val       = __fmul_rn(val, __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-val))));
        out[i]    = __fmul_rn(val, u[i]);
    }
}

// ---------------------------------------------------------------- embedding rows -> F32 (ggml_wrapper.cpp:181-211)
// dequantize_row_q4_0 / q8_0 / q4_K / q6_K (ggml-quants.c:1536, 1630, 2569, 2991) from the SoA repack.
struct WView { int dtype; int64_t K, N; const uint8_t *qs, *aux, *qh, *sc; };
__global__ void get_rows_kernel(WView w, const int32_t *tokens, int n, float *out) {
    const int64_t K = w.K;
    const int64_t total = (int64_t)n * K;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = o % K, i = o / K, row = tokens[i];
        float v;
        if (w.dtype == PS_F32) {
            v = ((const float *)w.qs)[row * K + e];
        } else if (w.dtype == PS_Q4_0) {
            const int64_t nu = (K + 127) / 128, g = row / 16, blk = e / 32;
            const int r = (int)(row % 16), j = (int)(e % 32), byte = j & 15, bl = (int)(blk & 3);
            const int64_t gs = g * nu + blk / 4;
            const uint8_t bv = w.qs[((gs * 16 + r) * 4 + (byte >> 2)) * 16 + bl * 4 + (byte & 3)];
            const int q = (j < 16 ? (bv & 0xF) : (bv >> 4)) - 8;
            v = __fmul_rn((float)q, ps_h2f(((const uint16_t *)w.aux)[(gs * 16 + r) * 4 + bl]));
        } else if (w.dtype == PS_Q8_0) {
            const int64_t nu = (K + 127) / 128, g = row / 8, blk = e / 32;
            const int r = (int)(row % 8), j = (int)(e % 32), bl = (int)(blk & 3);
            const int64_t gs = g * nu + blk / 4;
            const int8_t qv = ((const int8_t *)w.qs)[((gs * 8 + r) * 8 + (j >> 2)) * 16 + bl * 4 + (j & 3)];
            v = __fmul_rn((float)qv, ps_h2f(((const uint16_t *)w.aux)[(gs * 8 + r) * 4 + bl]));
        } else if (w.dtype == PS_Q4_K) {
            const int64_t nsb = K / 256, sb = e / 256, g = row / 8;
            const int r = (int)(row % 8), rr = (int)(e % 256), j = rr / 64, l = rr % 64, byte = l & 31;
            const int64_t gs = g * nsb + sb;
            const uint4 h = ((const uint4 *)w.aux)[gs * 8 + r];
            int sc, m; ps_scale_min_k4(2 * j + (l >= 32), h.y, h.z, h.w, sc, m);
            const uint8_t bv = w.qs[((gs * 8 + r) * 8 + (byte >> 2)) * 16 + j * 4 + (byte & 3)];
            const int q = (l < 32) ? (bv & 0xF) : (bv >> 4);
            const float d = ps_h2f((uint16_t)(h.x & 0xffff)), mn = ps_h2f((uint16_t)(h.x >> 16));
            v = __fsub_rn(__fmul_rn(__fmul_rn(d, (float)sc), (float)q), __fmul_rn(mn, (float)m));
        } else if (w.dtype == PS_Q5_K) { // dequantize_row_q5_K (ggml-quants.c:2777-2802): d1 * (q + 16 h) - m1
            const int64_t sb = e / 256; const int r = (int)(e % 256), j = r / 32, l = r % 32, u5 = l >> 2;
            const uint4 h = ((const uint4 *)w.sc)[row * (K / 256) + sb];
            int sc, m; ps_scale_min_k4(j, h.y, h.z, h.w, sc, m);
            const uint8_t bv = w.qs[row * (K / 2) + sb * 128 + (u5 * 4 + (j >> 1)) * 4 + (l & 3)];
            const int hb = (w.qh[row * (K / 8) + sb * 32 + u5 * 4 + (l & 3)] >> j) & 1;
            const int q = ((j & 1) ? (bv >> 4) : (bv & 0xF)) + 16 * hb;
            const float d = ps_h2f((uint16_t)(h.x & 0xffff)), mn = ps_h2f((uint16_t)(h.x >> 16));
            v = __fsub_rn(__fmul_rn(__fmul_rn(d, (float)sc), (float)q), __fmul_rn(mn, (float)m));
        } else { // Q6_K
            const int64_t sb = e / 256; const int r = (int)(e % 256), half = r / 128, rr = r % 128, sub = rr / 32, l = rr % 32;
            const uint8_t *ql = w.qs + row * (K / 2) + sb * 128; // lane-major inside the super-block (ps_internal.h)
            const uint8_t *qh = w.qh + row * (K / 4) + sb * 64;
            const int8_t *sc  = (const int8_t *)w.sc + row * (K / 16) + sb * 16 + half * 8;
            const int u6 = l >> 2;
            int lo = ql[(u6 * 4 + 2 * half + (sub & 1)) * 4 + (l & 3)];
            lo     = (sub >= 2) ? (lo >> 4) : (lo & 0xF);
            const int q = (int)(int8_t)(lo | (((qh[(u6 * 2 + half) * 4 + (l & 3)] >> (2 * sub)) & 3) << 4)) - 32;
            const float d = ps_h2f(((const uint16_t *)w.aux)[row * (K / 256) + sb]);
            v = __fmul_rn(__fmul_rn(d, (float)sc[l / 16 + 2 * sub]), (float)q);
        }
        out[o] = v;
    }
}

// ---------------------------------------------------------------- causal / tree mask (executor.cpp:210-224)
__global__ void get_mask_kernel(float *out, int64_t n_kv, int bs, const int32_t *pos, const uint8_t *tree) {
    const int64_t total = n_kv * bs;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = o % n_kv; const int i = (int)(o / n_kv);
        bool ok;
        if (tree) {
            const int64_t first = n_kv - bs; // batch tokens occupy the last bs cache slots
            ok = (j < first) ? true : tree[i * bs + (j - first)] != 0;
        } else {
            ok = j <= pos[i];
        }
        out[o] = ok ? 0.f : -INFINITY;
    }
}

// ---------------------------------------------------------------- arg-max, first maximum (prob_array.cpp:65-67)
__global__ __launch_bounds__(256) void argmax_kernel(const float *src, int64_t n, int32_t *out) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *x = src + (int64_t)blockIdx.x * n;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float v = x[i];
        if (v > best || (v == best && (int)i < idx)) { best = v; idx = (int)i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx; // a row of NaNs: std::max_element returns the first element
    }
}

inline unsigned grid1d(int64_t n, int bs = 256, int64_t cap = 4096) {
    int64_t g = (n + bs - 1) / bs;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

} // namespace

// ---- launchers (declared in ps_ops.h)
#include "ps_ops.h"

void psl_mul_mat_f32(hipStream_t st, const ps_tensor *dst, const ps_tensor *a, const ps_tensor *b) {
    const int64_t n_out = dst->ne[0] * dst->ne[1] * dst->ne[2] * dst->ne[3];
    hipLaunchKernelGGL(mul_mat_f32_kernel, dim3(grid1d(n_out, 8, 8192)), dim3(256), 0, st, tdesc(dst), tdesc(a), tdesc(b));
}
void psl_rms_norm(hipStream_t st, const ps_tensor *dst, const ps_tensor *src, const float *w, float eps) {
    const int64_t rows = src->ne[1] * src->ne[2] * src->ne[3];
    hipLaunchKernelGGL(rms_norm_kernel, dim3((unsigned)rows), dim3(256), 0, st, tdesc(dst), tdesc(src), w, eps);
}
void psl_rope(hipStream_t st, const ps_tensor *dst, const ps_tensor *src, const float *cache, int n_dims, int neox) {
    const int64_t total = src->ne[0] / 2 * src->ne[1] * src->ne[2] * src->ne[3];
    hipLaunchKernelGGL(rope_kernel, dim3(grid1d(total)), dim3(256), 0, st, tdesc(dst), tdesc(src), cache, n_dims, neox);
}
void psl_softmax_ext(hipStream_t st, const ps_tensor *dst, const ps_tensor *src, const float *mask, float scale) {
    const int64_t rows = src->ne[1] * src->ne[2] * src->ne[3];
    static unsigned long long attr = 0; // devices that have the attribute (rows of more than 64 KiB)
    if (ps_first_on_device(&attr)) (void)hipFuncSetAttribute((const void *)softmax_ext_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    hipLaunchKernelGGL(softmax_ext_kernel, dim3((unsigned)rows), dim3(256), (size_t)src->ne[0] * 4, st, tdesc(dst),
                       tdesc(src), mask, scale);
}
void psl_add(hipStream_t st, const ps_tensor *dst, const ps_tensor *a, const ps_tensor *b) {
    const int64_t total = dst->ne[0] * dst->ne[1] * dst->ne[2] * dst->ne[3];
    hipLaunchKernelGGL(add_kernel, dim3(grid1d(total)), dim3(256), 0, st, tdesc(dst), tdesc(a), tdesc(b));
}
void psl_dup(hipStream_t st, const ps_tensor *dst, const ps_tensor *src) {
    const int64_t total = src->ne[0] * src->ne[1] * src->ne[2] * src->ne[3];
    hipLaunchKernelGGL(dup_kernel, dim3(grid1d(total)), dim3(256), 0, st, tdesc(dst), tdesc(src));
}
void psl_silu_hadamard(hipStream_t st, float *out, const float *g, const float *u, int64_t n) {
    hipLaunchKernelGGL(silu_hadamard_kernel, dim3(grid1d(n)), dim3(256), 0, st, out, g, u, n);
}
void psl_get_rows(hipStream_t st, const ps_weight *w, const int32_t *tokens_dev, int n, float *out) {
    WView v{w->dtype, w->K, w->N, w->qs, w->aux, w->qh, w->sc};
    hipLaunchKernelGGL(get_rows_kernel, dim3(grid1d((int64_t)n * w->K)), dim3(256), 0, st, v, tokens_dev, n, out);
}
void psl_get_mask(hipStream_t st, float *out, int64_t n_kv, int bs, const int32_t *pos_dev, const uint8_t *tree_dev) {
    hipLaunchKernelGGL(get_mask_kernel, dim3(grid1d(n_kv * bs)), dim3(256), 0, st, out, n_kv, bs, pos_dev, tree_dev);
}
void psl_argmax(hipStream_t st, const float *src, int64_t n, int64_t rows, int32_t *out) {
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)rows), dim3(256), 0, st, src, n, out);
}
