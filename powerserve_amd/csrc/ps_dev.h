// Device-side helpers shared by the gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PS_WAVE 64

// ---- fp16 <-> fp32 (IEEE, RNE; matches F16C _cvtss_sh / _cvtsh_ss used by GGML_FP32_TO_FP16)
__device__ __forceinline__ float ps_h2f(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ uint16_t ps_f2h(float f) {
    return __half_as_ushort(__float2half_rn(f));
}

// ---- wave-level reductions (all 64 lanes end up with the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// reduce within aligned groups of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G>
__device__ __forceinline__ int group_sum_i(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- int8 dot: 4 signed bytes x 4 signed bytes + acc  (v_dot4_i32_i8)
__device__ __forceinline__ int dot4(int a, int b, int acc) {
    return __builtin_amdgcn_sdot4(a, b, acc, false);
}

// ---- streaming (read-once) 16-byte load: non-temporal so weight bytes do not displace L2-resident
//      activations (MI355X guide: nt-weights, -18 % issue->landed latency on decode weight streams)
typedef uint32_t ps_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream16(const void *p) {
    const ps_u32x4 v = __builtin_nontemporal_load((const ps_u32x4 *)p); // global_load_dwordx4 ... nt
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- ggml_v_expf (AVX2+FMA variant, libs/ggml/src/ggml.c:2685-2723), one lane.  Same operation
//      sequence with explicit fmaf so softmax matches the reference to the last bit on the vector part.
__device__ __forceinline__ float ps_v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = __fmaf_rn(x, 0x1.715476p+0f, r);
    const float n = __fsub_rn(z, r);
    const float b = __fmaf_rn(-n, 0x1.7f7d1cp-20f, __fmaf_rn(-n, 0x1.62e4p-1f, x));
    const uint32_t e = __float_as_uint(z) << 23;
    const float k = __uint_as_float(e + 0x3f800000u);
    const bool c = fabsf(n) > 126.0f;
    const float u = __fmul_rn(b, b);
    const float j = __fmaf_rn(__fmaf_rn(__fmaf_rn(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u,
                                        __fmaf_rn(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)),
                              u, __fmul_rn(0x1.ffffecp-1f, b));
    if (!c) return __fmaf_rn(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = __uint_as_float(g + 0x7f000000u);
    const float s2 = __uint_as_float(e - g);
    if (fabsf(n) > 192.0f) return __fmul_rn(s1, s1);
    return __fmul_rn(__fmaf_rn(s2, j, s2), s1);
}

// ---- get_scale_min_k4 (libs/ggml/src/ggml-quants.c:1912-1920) on the 12 scale bytes held as 3 dwords
__device__ __forceinline__ void ps_scale_min_k4(int is, uint32_t s0, uint32_t s1, uint32_t s2, int &sc, int &m) {
    const int k = is & 3, sh = 8 * k;
    const uint32_t a = (s0 >> sh) & 0xff, b = (s1 >> sh) & 0xff, c = (s2 >> sh) & 0xff;
    if (is < 4) {
        sc = a & 63;
        m  = b & 63;
    } else {
        sc = (c & 0xF) | ((a >> 6) << 4);
        m  = (c >> 4) | ((b >> 6) << 4);
    }
}
