// KV-cache attention for the fused model path (gfx950, HBM-bound on the FP32 K/V stream).
//
// Fuses what NormAttention::build emits after the QKV mat-muls
// (src/model/module/norm_attention.cpp:72-151): ROPE x2, TRANSPOSE, VIEW+COPY x2 (KV append),
// PERMUTE, K-view MAT_MUL, GET_MASK (src/executor/executor.cpp:210-224), SOFTMAX_EXT, V-view MAT_MUL,
// PERMUTE+CONT — 13 graph ops — into three launches with no intermediate layout shuffles:
//   rope_append      : rotate q in place, rotate k into its K-cache row, scatter v into the transposed V cache
//   attn_scores      : raw q·K for a 64-position tile per workgroup (all q heads of one kv head share the K tile)
//   attn_softmax_pv  : scale + mask + softmax (ggml_v_expf polynomial, double row sum) held in LDS, then V·p
// KV layout is the reference's: K [n_ctx][kv_dim], V [kv_dim][n_ctx] FP32 (backend/ggml/ggml_kv_cache.cpp:48-57).
// All position-dependent values come from a device-resident ps_step_state so a captured hipGraph replays.
#include "ps_dev.h"
#include "ps_ops.h"

namespace {

// ---------------------------------------------------------------- rope + KV append
__global__ void rope_append_kernel(psl_attn_args a, int bs) {
    const int hs = a.head_size, dim = a.n_heads * hs, kvd = a.n_kv_heads * hs, half = a.n_dims / 2;
    const int pos0 = a.state->pos0;
    const int64_t nq = (int64_t)bs * a.n_heads * (hs / 2), nk = (int64_t)bs * a.n_kv_heads * (hs / 2), nv = (int64_t)bs * kvd;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nq + nk + nv; o += (int64_t)gridDim.x * blockDim.x) {
        if (o < nq + nk) {
            const bool isq = o < nq;
            const int64_t oo = isq ? o : o - nq;
            const int nh = isq ? a.n_heads : a.n_kv_heads;
            const int pi = (int)(oo % (hs / 2));
            const int h = (int)((oo / (hs / 2)) % nh), i = (int)(oo / ((int64_t)(hs / 2) * nh));
            const int p = pos0 + i, i0 = 2 * pi;
            const float *src = isq ? a.q + (int64_t)i * dim + h * hs : a.k + (int64_t)i * kvd + h * hs;
            float *dst = isq ? a.q + (int64_t)i * dim + h * hs : a.k_cache + (int64_t)p * kvd + h * hs;
            if (i0 >= a.n_dims) { dst[i0] = src[i0]; dst[i0 + 1] = src[i0 + 1]; continue; }
            const float c = a.rope_table[(int64_t)p * hs + i0], s = a.rope_table[(int64_t)p * hs + i0 + 1];
            const int ia = a.neox ? pi : i0, ib = a.neox ? pi + half : i0 + 1;
            const float x0 = src[ia], x1 = src[ib];
            dst[ia] = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s));
            dst[ib] = __fadd_rn(__fmul_rn(x0, s), __fmul_rn(x1, c));
        } else {
            const int64_t oo = o - nq - nk;
            const int d = (int)(oo % kvd), i = (int)(oo / kvd);
            a.v_cache[(int64_t)d * a.n_ctx + pos0 + i] = a.v[(int64_t)i * kvd + d];
        }
    }
}

// ---------------------------------------------------------------- scores: s[i][h][j] = q[i][h] · K[j][kvh]
// grid (n_ctx/64, n_kv_heads, bs); 4 waves; a wave takes 8 K rows at a time, 8 lanes per row, each lane
// NV float4 of the row (coalesced 128 B per 8 lanes); the K fragment stays in registers while the q
// fragments of the r2 heads sharing this kv head stream from L1.
template <int NV>
__global__ __launch_bounds__(256) void attn_scores_kernel(psl_attn_args a) {
    const int hs = NV * 32, dim = a.n_heads * hs, kvd = a.n_kv_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 7, rowl = lane >> 3;
    const int kvh = blockIdx.y, i = blockIdx.z;
    const int n_kv = a.state->pos0 + a.state->bs;
    const int j0 = blockIdx.x * 64;
    if (j0 >= n_kv) return;
    const float *qb = a.q + (int64_t)i * dim + (int64_t)kvh * r2 * hs;
    float *sb       = a.scores + ((int64_t)i * a.n_heads + (int64_t)kvh * r2) * a.n_ctx;
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int j = j0 + wave * 16 + rb * 8 + rowl;
        const bool live = j < n_kv;
        float4 kf[NV];
        const float *kr = a.k_cache + (int64_t)(live ? j : 0) * kvd + kvh * hs;
#pragma unroll
        for (int m = 0; m < NV; m++) kf[m] = *(const float4 *)(kr + m * 32 + sub * 4);
        for (int g = 0; g < r2; g++) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < NV; m++) {
                const float4 qf = *(const float4 *)(qb + g * hs + m * 32 + sub * 4);
                s = __fmaf_rn(kf[m].x, qf.x, s);
                s = __fmaf_rn(kf[m].y, qf.y, s);
                s = __fmaf_rn(kf[m].z, qf.z, s);
                s = __fmaf_rn(kf[m].w, qf.w, s);
            }
            s = group_sum<8>(s);
            if (live && sub == 0) sb[(int64_t)g * a.n_ctx + j] = s;
        }
    }
}

// ---------------------------------------------------------------- softmax + V·p
// grid (hs/4, n_kv_heads, bs); each wave produces one output channel d for the r2 heads of the group.
// LDS: p[r2][n_kv] floats.
constexpr int R2MAX = 8;
__global__ __launch_bounds__(256) void attn_softmax_pv_kernel(psl_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float p[];
    __shared__ float redf[4];
    __shared__ double redd[4];
    const int hs = a.head_size, dim = a.n_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kvh = blockIdx.y, i = blockIdx.z, bs = a.state->bs, pos0 = a.state->pos0;
    const int n_kv = pos0 + bs;
    const int n_kv4 = (n_kv + 3) & ~3;
    // ---- softmax rows of the r2 heads (ggml.c:14889-14925, ggml_vec_soft_max_f32 :2814-2863)
    for (int g = 0; g < r2; g++) {
        const float *sp = a.scores + ((int64_t)i * a.n_heads + (int64_t)kvh * r2 + g) * a.n_ctx;
        float *pg = p + (int64_t)g * n_kv4;
        float mx = -INFINITY;
        for (int j = threadIdx.x; j < n_kv4; j += 256) {
            float v = -INFINITY;
            if (j < n_kv) {
                bool ok = (j < pos0) ? true : (a.tree ? a.tree[i * bs + (j - pos0)] != 0 : (j - pos0) <= i);
                v = __fmul_rn(sp[j], a.scale);
                v = __fadd_rn(v, ok ? 0.f : -INFINITY);
            }
            pg[j] = v;
            mx = fmaxf(mx, v);
        }
        mx = wave_max(mx);
        __syncthreads(); // protects redf/redd reuse across g
        if (lane == 0) redf[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        double sum = 0.0;
        for (int j = threadIdx.x; j < n_kv4; j += 256) {
            const float e = (j < n_kv) ? ps_v_expf(__fsub_rn(pg[j], mx)) : 0.f;
            pg[j] = e;
            sum += (double)e;
        }
        sum = wave_sum_d(sum);
        if (lane == 0) redd[wave] = sum;
        __syncthreads();
        const double tot = (redd[0] + redd[1]) + (redd[2] + redd[3]);
        const float inv = (float)(1.0 / tot);
        for (int j = threadIdx.x; j < n_kv4; j += 256) pg[j] = __fmul_rn(pg[j], inv);
    }
    __syncthreads();
    // ---- V·p: channel d = blockIdx.x*4 + wave of kv head kvh; V row is contiguous along positions
    const int d = blockIdx.x * 4 + wave;
    const float *vr = a.v_cache + ((int64_t)kvh * hs + d) * a.n_ctx;
    float acc[R2MAX];
#pragma unroll
    for (int g = 0; g < R2MAX; g++) acc[g] = 0.f;
    for (int j = lane * 4; j < n_kv4; j += 256) {
        float4 v = *(const float4 *)(vr + j); // n_ctx is a multiple of 4; slots >= n_kv hold p == 0
        if (j + 3 >= n_kv) { // never multiply stale cache contents (could be NaN) by 0
            if (j + 0 >= n_kv) v.x = 0.f;
            if (j + 1 >= n_kv) v.y = 0.f;
            if (j + 2 >= n_kv) v.z = 0.f;
            if (j + 3 >= n_kv) v.w = 0.f;
        }
#pragma unroll
        for (int g = 0; g < R2MAX; g++) {
            if (g < r2) {
                const float4 pv = *(const float4 *)(p + (int64_t)g * n_kv4 + j);
                acc[g] = __fmaf_rn(v.x, pv.x, acc[g]);
                acc[g] = __fmaf_rn(v.y, pv.y, acc[g]);
                acc[g] = __fmaf_rn(v.z, pv.z, acc[g]);
                acc[g] = __fmaf_rn(v.w, pv.w, acc[g]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < R2MAX; g++) {
        if (g < r2) {
            const float s = wave_sum(acc[g]);
            if (lane == 0) a.att[(int64_t)i * dim + ((int64_t)kvh * r2 + g) * hs + d] = s;
        }
    }
}

} // namespace

void psl_rope_append(hipStream_t st, const psl_attn_args &a, int bs) {
    const int64_t n = (int64_t)bs * (a.n_heads + a.n_kv_heads) * (a.head_size / 2) + (int64_t)bs * a.n_kv_heads * a.head_size;
    int64_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(rope_append_kernel, dim3((unsigned)g), dim3(256), 0, st, a, bs);
}

void psl_attn_scores(hipStream_t st, const psl_attn_args &a, int bs) {
    dim3 g((unsigned)((a.n_ctx + 63) / 64), (unsigned)a.n_kv_heads, (unsigned)bs);
    if (a.head_size == 128) hipLaunchKernelGGL(attn_scores_kernel<4>, g, dim3(256), 0, st, a);
    else if (a.head_size == 64) hipLaunchKernelGGL(attn_scores_kernel<2>, g, dim3(256), 0, st, a);
    else if (a.head_size == 32) hipLaunchKernelGGL(attn_scores_kernel<1>, g, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_scores_kernel<3>, g, dim3(256), 0, st, a); // head_size 96
}

void psl_attn_softmax_pv(hipStream_t st, const psl_attn_args &a, int bs) {
    const int r2 = a.n_heads / a.n_kv_heads;
    const size_t smem = (size_t)r2 * (size_t)((a.n_ctx + 3) & ~3) * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)attn_softmax_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        attr = true;
    }
    dim3 g((unsigned)(a.head_size / 4), (unsigned)a.n_kv_heads, (unsigned)bs);
    hipLaunchKernelGGL(attn_softmax_pv_kernel, g, dim3(256), smem, st, a);
}
