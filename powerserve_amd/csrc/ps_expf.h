// expf with the exact operation sequence of glibc's expf (sysdeps/ieee754/flt-32/e_expf.c, the ARM
// "optimized routines" algorithm, glibc >= 2.27): double-precision table-driven evaluation, result rounded
// once to float.  glibc is the only arithmetic on the reference's path that lives outside /root/reference
// (libm expf in GGMLBackend::silu_hadamard, src/backend/ggml/ggml.cpp:125, and the scalar tail of
// ggml_vec_soft_max_f32, libs/ggml/src/ggml.c:2856-2860); its algorithm and constants are public.
// x86-64 glibc dispatches to the FMA build of this routine (__expf_fma) on every FMA-capable CPU, so the
// three multiply-adds below are fused.  tests/test_expf.py checks this header bit-for-bit against the
// host's libm on millions of inputs.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PS_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define PS_HD static inline
#endif

#define PS_EXP2F_N 32
#if defined(__HIPCC__)
__device__ __constant__
#endif
static const uint64_t ps_exp2f_tab[PS_EXP2F_N] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

#if defined(__HIPCC__)
__device__ __forceinline__ float ps_expf_glibc(float x, const uint64_t *tab = ps_exp2f_tab) {
#else
static inline float ps_expf_glibc(float x) {
    const uint64_t *tab = ps_exp2f_tab;
#endif
    uint32_t ix;
    memcpy(&ix, &x, 4);
    const uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop >= 0x42b) { // |x| >= 88 or nan
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_inff();
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    const double xd      = (double)x;
    const double InvLn2N = 0x1.71547652b82fep+0 * PS_EXP2F_N;
    const double SHIFT   = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / PS_EXP2F_N / PS_EXP2F_N / PS_EXP2F_N;
    const double C1 = 0x1.ebfce50fac4f3p-3 / PS_EXP2F_N / PS_EXP2F_N;
    const double C2 = 0x1.62e42ff0c52d6p-1 / PS_EXP2F_N;
    double z  = InvLn2N * xd;
    double kd = z + SHIFT;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    const double r = z - kd;
    uint64_t t     = tab[ki % PS_EXP2F_N];
    t += ki << (52 - 5);
    double s;
    memcpy(&s, &t, 8);
    z               = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y        = __builtin_fma(C2, r, 1.0);
    y               = __builtin_fma(z, r2, y);
    y               = y * s;
    return (float)y;
}
