// Workgroup-wide activation-row quantizer shared by the stand-alone kernel (k_quant.hip) and the fused
// GEMV prologue (k_gemv.hip).  Bit-exact restatement of quantize_row_q8_0 (AVX2 branch,
// libs/ggml/src/ggml-quants.c:957-1039) and quantize_row_q8_K (:3799-3835), with optional producers:
//   MODE 1  RMSNorm  y = x * (w * scale)                  (ggml.c:12667-12720, :2442-2470)
//   MODE 2  SiLU*up  val = g * (1/(1+expf(-g))) * u        (src/backend/ggml/ggml.cpp:115-129)
// 256 threads; a wave owns tiles of 256 consecutive elements, lane l elements 4l..4l+3.  Output pointers may
// be global or LDS.  `red` is a __shared__ double[4] scratch.  Ends with __syncthreads().
#pragma once
#include "ps_dev.h"
#include "ps_expf.h"

__device__ __forceinline__ float ps_silu_mul(float g, float u) {
    float val = g;
    val       = __fmul_rn(val, __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-val))));
    return __fmul_rn(val, u);
}

template <int VDT, int MODE>
__device__ __forceinline__ void ps_quantize_row_wg(const float *x, const float *x2, const float *w, float eps, int64_t K,
                                                   int8_t *qs, float *d, int16_t *bs16, double *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_tiles = (K + 255) / 256;
    float scale = 1.0f;
    if (MODE == 1) {
        // sum += (double)(x*x); mean = sum/ne00; scale = 1/sqrtf(mean + eps)
        double s = 0.0;
        for (int64_t t = wave; t < n_tiles; t += 4) {
            const int64_t e = t * 256 + lane * 4;
            if (e < K) {
                const float4 v = *(const float4 *)(x + e);
                s += (double)__fmul_rn(v.x, v.x);
                s += (double)__fmul_rn(v.y, v.y);
                s += (double)__fmul_rn(v.z, v.z);
                s += (double)__fmul_rn(v.w, v.w);
            }
        }
        s = wave_sum_d(s);
        __syncthreads();
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const double tot = (red[0] + red[1]) + (red[2] + red[3]);
        const float mean = (float)(tot / (double)K);
        scale            = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, eps)));
    }
    for (int64_t t = wave; t < n_tiles; t += 4) {
        const int64_t e = t * 256 + lane * 4;
        const bool live = e < K;
        float v[4]      = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const float4 xv = *(const float4 *)(x + e);
            if (MODE == 1) {
                const float4 wv = *(const float4 *)(w + e);
                v[0] = __fmul_rn(xv.x, __fmul_rn(wv.x, scale));
                v[1] = __fmul_rn(xv.y, __fmul_rn(wv.y, scale));
                v[2] = __fmul_rn(xv.z, __fmul_rn(wv.z, scale));
                v[3] = __fmul_rn(xv.w, __fmul_rn(wv.w, scale));
            } else if (MODE == 2) {
                const float4 uv = *(const float4 *)(x2 + e);
                v[0] = ps_silu_mul(xv.x, uv.x);
                v[1] = ps_silu_mul(xv.y, uv.y);
                v[2] = ps_silu_mul(xv.z, uv.z);
                v[3] = ps_silu_mul(xv.w, uv.w);
            } else {
                v[0] = xv.x; v[1] = xv.y; v[2] = xv.z; v[3] = xv.w;
            }
        }
        int q[4];
        if (VDT == PS_Q8_0) {
            float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            amax       = group_max<8>(amax);
            const float dd = __fdiv_rn(amax, 127.f);
            const float id = (amax != 0.0f) ? __fdiv_rn(127.f, amax) : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = __float2int_rn(__fmul_rn(v[i], id)); // round-half-even
            if (live && (lane & 7) == 0) d[e / 32] = ps_h2f(ps_f2h(dd));
        } else { // Q8_K: the first element (index order) with the strictly largest |x| decides the sign
            float amax = 0.f, mx = 0.f;
            int idx    = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float ax = fabsf(v[i]);
                if (ax > amax) { amax = ax; mx = v[i]; idx = lane * 4 + i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float oa = __shfl_xor(amax, o, 64), om = __shfl_xor(mx, o, 64);
                const int oi   = __shfl_xor(idx, o, 64);
                if (oa > amax || (oa == amax && oi < idx)) { amax = oa; mx = om; idx = oi; }
            }
            if (amax == 0.f) {
                q[0] = q[1] = q[2] = q[3] = 0;
                if (live && lane == 0) d[t] = 0.f;
            } else {
                const float iscale = __fdiv_rn(-127.f, mx);
#pragma unroll
                for (int i = 0; i < 4; i++) q[i] = min(127, __float2int_rn(__fmul_rn(iscale, v[i])));
                if (live && lane == 0) d[t] = __fdiv_rn(1.0f, iscale);
            }
        }
        const int s16 = group_sum_i<4>(q[0] + q[1] + q[2] + q[3]);
        if (live) {
            const uint32_t packed = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) |
                                    ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
            *(uint32_t *)(qs + e) = packed;
            if ((lane & 3) == 0) bs16[e / 16] = (int16_t)s16;
        }
    }
    __syncthreads();
}
