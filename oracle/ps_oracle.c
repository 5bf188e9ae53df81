/* oracle/ps_oracle.c — TEST INFRASTRUCTURE ONLY.  NOT part of the product path.
 *
 * Plain-C restatement of the arithmetic on PowerServe's ggml decode hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / reported CPU baseline.  The product (powerserve_amd/) never links or calls this file and
 * fails loudly when its HIP library is missing.
 *
 * PINNING: every function below is checked (tests/test_oracle_vs_ref.py, dev container only) against
 * the real reference compiled from /root/reference into oracle/_ref/libps_ref.so (oracle/Makefile),
 * and against golden vectors that library produced (tests/golden/, generator oracle/gen_golden.py).
 * The reference's own tests hold no known-answer vectors for this path.
 *
 * Floating-point conventions: the reference's x86 build takes ggml's AVX2 code paths (8 fp32 lanes,
 * explicit FMA intrinsics).  The dot products below reproduce that lane structure with scalar fmaf(),
 * so on an AVX2 reference build they agree BIT-FOR-BIT; scalar C parts follow the C source as written
 * (no FMA contraction: both this file and oracle/_ref are built with -ffp-contract=off).
 *
 * All file:line citations are relative to /root/reference/.
 */
#define _GNU_SOURCE
#include "ps_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define QK 32
#define QK_K 256

/* ------------------------------------------------------------------ the reference's SECOND build (round 5)
 * The reference's own CMake sets no floating-point contraction flag (CMakeLists.txt:24-33, libs/ggml/src/CMakeLists.txt:1173), so a
 * stock build on an FMA machine is GCC's default -ffp-contract=fast.  Of everything on the hot path exactly THREE places come out
 * different (every function of both builds compared by FMA-instruction count, then op by op: tests/test_ref_fast.py): the RoPE rotation
 * (ggml.c:15455-15456, :15474-15475), the scalar leftovers of ggml_vec_dot_f32 (ggml.c:2123-2125: V.p with n_kv % 32 != 0) and Q5_K's
 * `summs += dmin * hsum` (ggml-quants.c:8411).  pso_set_contract(1) makes this file follow THAT build (GCC 11.4 -O3 -mavx2 -mfma, the
 * dev container's compiler: which operand pair is fused, and that the leftover loop is vectorised by 8 and by 4 WITHOUT fusing and only
 * its last n % 4 steps are scalar fmas, is that compiler's choice) -- pinned bit for bit against oracle/_ref/libps_ref_fast.so.
 * Default 0 = oracle/_ref/libps_ref.so (-ffp-contract=off): what the golden vectors, the HIP library and "bit-exact" refer to. */
static int g_contract = 0;
void pso_set_contract(int on) { g_contract = on; }
int pso_get_contract(void) { return g_contract; }

/* ------------------------------------------------------------------ block layouts
 * libs/ggml/src/ggml-common.h:158-162 (q4_0), :200-204 (q8_0), :296-310 (q4_K), :317-328 (q5_K), :335-340 (q6_K),
 * :344-348 (q8_K). */
#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; } blk_q4_0;                               /* 18 B */
typedef struct { uint16_t d; int8_t qs[32]; } blk_q8_0;                                /* 34 B */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; } blk_q4_K;    /* 144 B */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; } blk_q5_K; /* 176 B */
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; } blk_q6_K; /* 210 B */
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; } blk_q8_K;               /* 292 B */
#pragma pack(pop)

/* ------------------------------------------------------------------ fp16 (IEEE, RNE == F16C) */
float pso_fp16_to_fp32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, bits;
    if (e == 0) {
        if (m == 0) {
            bits = s;
        } else { /* subnormal */
            int sh = 0;
            while (!(m & 0x400)) { m <<= 1; sh++; }
            m &= 0x3ff;
            bits = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        bits = s | 0x7f800000u | (m << 13);
    } else {
        bits = s | ((e + 112) << 23) | (m << 13);
    }
    float f; memcpy(&f, &bits, 4); return f;
}

uint16_t pso_fp32_to_fp16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t s = (x >> 16) & 0x8000; int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15; uint32_t m = x & 0x7fffff;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(s | 0x7c00 | (m ? 0x200 | (m >> 13) : 0));
    if (e >= 31) return (uint16_t)(s | 0x7c00);
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000; int sh = 14 - e; uint32_t hm = m >> sh, rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(s | hm);
    }
    uint32_t hm = m >> 13, rem = m & 0x1fff; uint16_t r = (uint16_t)(s | ((uint32_t)e << 10) | hm);
    if (rem > 0x1000 || (rem == 0x1000 && (hm & 1))) r++;
    return r;
}

/* ------------------------------------------------------------------ type traits (ggml.c:681-1015) */
size_t pso_type_size(int t) {
    switch (t) {
    case PSO_F32: return 4; case PSO_F16: return 2; case PSO_I32: return 4;
    case PSO_Q4_0: return sizeof(blk_q4_0); case PSO_Q8_0: return sizeof(blk_q8_0);
    case PSO_Q4_K: return sizeof(blk_q4_K); case PSO_Q5_K: return sizeof(blk_q5_K); case PSO_Q6_K: return sizeof(blk_q6_K);
    case PSO_Q8_K: return sizeof(blk_q8_K);
    }
    return 0;
}
int64_t pso_blck_size(int t) {
    switch (t) {
    case PSO_Q4_0: case PSO_Q8_0: return QK;
    case PSO_Q4_K: case PSO_Q5_K: case PSO_Q6_K: case PSO_Q8_K: return QK_K;
    }
    return 1;
}
size_t pso_row_size(int t, int64_t k) { return pso_type_size(t) * (size_t)(k / pso_blck_size(t)); }
/* vec_dot_type: Q4_0,Q8_0 -> Q8_0 (ggml.c:734-748,814-830); Q4_K,Q5_K,Q6_K -> Q8_K (:865-900); F32 -> F32 */
int pso_vec_dot_type(int t) {
    switch (t) {
    case PSO_Q4_0: case PSO_Q8_0: return PSO_Q8_0;
    case PSO_Q4_K: case PSO_Q5_K: case PSO_Q6_K: return PSO_Q8_K;
    }
    return t;
}

/* ------------------------------------------------------------------ activation quantizers */
/* quantize_row_q8_0, AVX2 branch (ggml-quants.c:957-1039): amax, d = amax/127, id = 127/amax,
 * q = round-half-even(x*id); stored d is fp16.  (The scalar _ref variant :862-885 uses roundf and
 * 1/d and can differ in rare ties; x86 builds of the reference never take it on the mat-mul path.) */
void pso_quantize_row_q8_0(const float *x, void *vy, int64_t k) {
    blk_q8_0 *y = (blk_q8_0 *)vy;
    for (int64_t i = 0; i < k / QK; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) { float a = fabsf(x[i * QK + j]); if (a > amax) amax = a; }
        const float d  = amax / 127.f;
        y[i].d         = pso_fp32_to_fp16(d);
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        for (int j = 0; j < QK; j++) y[i].qs[j] = (int8_t)lrintf(x[i * QK + j] * id); /* RNE */
    }
}

/* nearest_int (ggml-quants.c:1653-1658) */
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f; int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

/* quantize_row_q8_K_ref (ggml-quants.c:3799-3835) */
void pso_quantize_row_q8_K(const float *x, void *vy, int64_t k) {
    blk_q8_K *y = (blk_q8_K *)vy;
    for (int64_t i = 0; i < k / QK_K; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < QK_K; ++j) { float ax = fabsf(x[j]); if (ax > amax) { amax = ax; max = x[j]; } }
        if (!amax) {
            y[i].d = 0; memset(y[i].qs, 0, QK_K);
            /* reference leaves bsums untouched here (uninitialised workspace); they are multiplied by
             * d == 0 in every consumer, so any finite value is equivalent.  We define them as 0. */
            memset(y[i].bsums, 0, sizeof(y[i].bsums));
            x += QK_K; continue;
        }
        const float iscale = -127.f / max;
        for (int j = 0; j < QK_K; ++j) { int v = nearest_int(iscale * x[j]); y[i].qs[j] = (int8_t)(v < 127 ? v : 127); }
        for (int j = 0; j < QK_K / 16; ++j) {
            int sum = 0; for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = (int16_t)sum;
        }
        y[i].d = 1 / iscale;
        x += QK_K;
    }
}

void pso_from_float(int vdt, const float *x, void *y, int64_t k) {
    if (vdt == PSO_Q8_0) pso_quantize_row_q8_0(x, y, k);
    else if (vdt == PSO_Q8_K) pso_quantize_row_q8_K(x, y, k);
    else memcpy(y, x, (size_t)k * 4);
}

/* ------------------------------------------------------------------ dequantizers */
/* get_scale_min_k4 (ggml-quants.c:1912-1920) */
static inline void get_scale_min_k4(int j, const uint8_t *q, uint8_t *d, uint8_t *m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

void pso_dequantize_row(int type, const void *vx, float *y, int64_t k) {
    if (type == PSO_F32) { memcpy(y, vx, (size_t)k * 4); return; }
    if (type == PSO_Q4_0) { /* dequantize_row_q4_0 (ggml-quants.c:1536-1554) */
        const blk_q4_0 *x = vx;
        for (int64_t i = 0; i < k / QK; i++) {
            const float d = pso_fp16_to_fp32(x[i].d);
            for (int j = 0; j < 16; ++j) {
                const int x0 = (x[i].qs[j] & 0x0F) - 8, x1 = (x[i].qs[j] >> 4) - 8;
                y[i * QK + j] = x0 * d; y[i * QK + j + 16] = x1 * d;
            }
        }
    } else if (type == PSO_Q8_0) { /* dequantize_row_q8_0 (ggml-quants.c:1630-1646) */
        const blk_q8_0 *x = vx;
        for (int64_t i = 0; i < k / QK; i++) {
            const float d = pso_fp16_to_fp32(x[i].d);
            for (int j = 0; j < QK; ++j) y[i * QK + j] = x[i].qs[j] * d;
        }
    } else if (type == PSO_Q4_K) { /* dequantize_row_q4_K (ggml-quants.c:2569-2590) */
        const blk_q4_K *x = vx;
        for (int64_t i = 0; i < k / QK_K; i++) {
            const uint8_t *q = x[i].qs;
            const float d = pso_fp16_to_fp32(x[i].d), min = pso_fp16_to_fp32(x[i].dmin);
            int is = 0; uint8_t sc, m;
            for (int j = 0; j < QK_K; j += 64) {
                get_scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d * sc, m1 = min * m;
                get_scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d * sc, m2 = min * m;
                for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
                for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
                q += 32; is += 2;
            }
        }
    } else if (type == PSO_Q5_K) { /* dequantize_row_q5_K (ggml-quants.c:2777-2802) */
        const blk_q5_K *x = vx;
        for (int64_t i = 0; i < k / QK_K; i++) {
            const uint8_t *ql = x[i].qs, *qh = x[i].qh;
            const float d = pso_fp16_to_fp32(x[i].d), min = pso_fp16_to_fp32(x[i].dmin);
            int is = 0; uint8_t sc, m, u1 = 1, u2 = 2;
            for (int j = 0; j < QK_K; j += 64) {
                get_scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d * sc, m1 = min * m;
                get_scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d * sc, m2 = min * m;
                for (int l = 0; l < 32; ++l) *y++ = d1 * ((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
                for (int l = 0; l < 32; ++l) *y++ = d2 * ((ql[l] >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
                ql += 32; is += 2; u1 <<= 2; u2 <<= 2;
            }
        }
    } else if (type == PSO_Q6_K) { /* dequantize_row_q6_K (ggml-quants.c:2991-3020) */
        const blk_q6_K *x = vx;
        for (int64_t i = 0; i < k / QK_K; i++) {
            const float d = pso_fp16_to_fp32(x[i].d);
            const uint8_t *ql = x[i].ql, *qh = x[i].qh; const int8_t *sc = x[i].scales;
            for (int n = 0; n < QK_K; n += 128) {
                for (int l = 0; l < 32; ++l) {
                    int is = l / 16;
                    const int8_t q1 = (int8_t)((ql[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                    const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                    const int8_t q3 = (int8_t)((ql[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                    const int8_t q4 = (int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                    y[l + 0] = d * sc[is + 0] * q1; y[l + 32] = d * sc[is + 2] * q2;
                    y[l + 64] = d * sc[is + 4] * q3; y[l + 96] = d * sc[is + 6] * q4;
                }
                y += 128; ql += 64; qh += 32; sc += 8;
            }
        }
    }
}

/* ------------------------------------------------------------------ dot products (AVX2 lane structure) */
/* hsum_float_8 (ggml-quants.c:62-68): (a0+a4)+(a2+a6) + (a1+a5)+(a3+a7) in that association */
static inline float hsum8(const float a[8]) {
    float r0 = a[4] + a[0], r1 = a[5] + a[1], r2 = a[6] + a[2], r3 = a[7] + a[3];
    r0 = r0 + r2; r1 = r1 + r3;
    return r0 + r1;
}

/* ggml_vec_dot_q4_0_q8_0, AVX2 branch (ggml-quants.c:4205-4228): lane u holds the 4-element partial
 * Σ_{e=4u..4u+3} (x_e-8)·y_e as float; acc[u] = fma(dx·dy, q[u], acc[u]). */
static float dot_q4_0_q8_0(int64_t n, const void *vx, const void *vy) {
    const blk_q4_0 *x = vx; const blk_q8_0 *y = vy; float acc[8] = {0};
    for (int64_t ib = 0; ib < n / QK; ++ib) {
        const float d = pso_fp16_to_fp32(x[ib].d) * pso_fp16_to_fp32(y[ib].d);
        int8_t xe[32];
        for (int j = 0; j < 16; j++) { xe[j] = (int8_t)((x[ib].qs[j] & 0xF) - 8); xe[j + 16] = (int8_t)((x[ib].qs[j] >> 4) - 8); }
        for (int u = 0; u < 8; u++) {
            int s = 0; for (int e = 4 * u; e < 4 * u + 4; e++) s += xe[e] * y[ib].qs[e];
            acc[u] = fmaf(d, (float)s, acc[u]);
        }
    }
    return hsum8(acc);
}

/* ggml_vec_dot_q8_0_q8_0, AVX2 branch (ggml-quants.c:5761-5782) */
static float dot_q8_0_q8_0(int64_t n, const void *vx, const void *vy) {
    const blk_q8_0 *x = vx; const blk_q8_0 *y = vy; float acc[8] = {0};
    for (int64_t ib = 0; ib < n / QK; ++ib) {
        const float d = pso_fp16_to_fp32(x[ib].d) * pso_fp16_to_fp32(y[ib].d);
        for (int u = 0; u < 8; u++) {
            int s = 0; for (int e = 4 * u; e < 4 * u + 4; e++) s += x[ib].qs[e] * y[ib].qs[e];
            acc[u] = fmaf(d, (float)s, acc[u]);
        }
    }
    return hsum8(acc);
}

/* ggml_vec_dot_q4_K_q8_K, AVX2 branch (ggml-quants.c:7809-7873) */
static float dot_q4_K_q8_K(int64_t n, const void *vx, const void *vy) {
    const blk_q4_K *x = vx; const blk_q8_K *y = vy; float acc[8] = {0}, acc_m[4] = {0};
    for (int64_t i = 0; i < n / QK_K; ++i) {
        const float d = y[i].d * pso_fp16_to_fp32(x[i].d);
        const float dmin = -y[i].d * pso_fp16_to_fp32(x[i].dmin);
        uint8_t sc[8], mn[8];
        for (int j = 0; j < 8; j++) get_scale_min_k4(j, x[i].scales, &sc[j], &mn[j]);
        /* q8s = hadd(bsums lo, bsums hi); prod[v] = mins[2v]*q8s[2v] + mins[2v+1]*q8s[2v+1] */
        int16_t q8s[8]; for (int t = 0; t < 8; t++) q8s[t] = (int16_t)(y[i].bsums[2 * t] + y[i].bsums[2 * t + 1]);
        for (int v = 0; v < 4; v++) {
            int prod = mn[2 * v] * q8s[2 * v] + mn[2 * v + 1] * q8s[2 * v + 1];
            acc_m[v] = fmaf(dmin, (float)prod, acc_m[v]);
        }
        int sumi[8] = {0};
        const uint8_t *q4 = x[i].qs; const int8_t *q8 = y[i].qs;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int u = 0; u < 8; u++) {
                int l = 0, h = 0;
                for (int e = 4 * u; e < 4 * u + 4; e++) { l += (q4[e] & 0xF) * q8[e]; h += (q4[e] >> 4) * q8[32 + e]; }
                sumi[u] += sc[2 * j] * l + sc[2 * j + 1] * h;
            }
            q4 += 32; q8 += 64;
        }
        for (int u = 0; u < 8; u++) acc[u] = fmaf(d, (float)sumi[u], acc[u]);
    }
    float m0 = acc_m[0] + acc_m[2], m1 = acc_m[1] + acc_m[3];
    return hsum8(acc) + (m0 + m1);
}

/* ggml_vec_dot_q5_K_q8_K, AVX2 branch (ggml-quants.c:8382-8459).  The integer part is as in q4_K with the fifth bit of
 * sub-vector b taken from bit b of qh; the mins are NOT kept in vector lanes here but in a scalar,
 *     summs += dmin * hsum(madd(mins, q8s))                                                   (:8411)
 * a float multiply followed by a float add in the C source.  Whether a compiler fuses the two is its fp-contraction
 * setting; the reference library this oracle is pinned to (oracle/_ref) is built with -ffp-contract=off, as is this file:
 * two roundings. */
static float dot_q5_K_q8_K(int64_t n, const void *vx, const void *vy) {
    const blk_q5_K *x = vx; const blk_q8_K *y = vy; float acc[8] = {0}; float summs = 0.f;
    for (int64_t i = 0; i < n / QK_K; ++i) {
        const float d = y[i].d * pso_fp16_to_fp32(x[i].d);
        const float dmin = -y[i].d * pso_fp16_to_fp32(x[i].dmin);
        uint8_t sc[8], mn[8];
        for (int j = 0; j < 8; j++) get_scale_min_k4(j, x[i].scales, &sc[j], &mn[j]);
        int hsum = 0; /* madd of the mins with the pairwise sums of bsums, then both hadds: one int32 */
        for (int t = 0; t < 8; t++) hsum += mn[t] * (int16_t)(y[i].bsums[2 * t] + y[i].bsums[2 * t + 1]);
        summs = g_contract ? fmaf(dmin, (float)hsum, summs) : summs + dmin * (float)hsum;
        int sumi[8] = {0};
        const uint8_t *q5 = x[i].qs, *qh = x[i].qh; const int8_t *q8 = y[i].qs;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int u = 0; u < 8; u++) {
                int l = 0, h = 0;
                for (int e = 4 * u; e < 4 * u + 4; e++) {
                    l += ((q5[e] & 0xF) + (((qh[e] >> (2 * j)) & 1) << 4)) * q8[e];
                    h += ((q5[e] >> 4) + (((qh[e] >> (2 * j + 1)) & 1) << 4)) * q8[32 + e];
                }
                sumi[u] += sc[2 * j] * l + sc[2 * j + 1] * h;
            }
            q5 += 32; q8 += 64;
        }
        for (int u = 0; u < 8; u++) acc[u] = fmaf(d, (float)sumi[u], acc[u]);
    }
    return hsum8(acc) + summs;
}

/* ggml_vec_dot_q6_K_q8_K, AVX2 branch (ggml-quants.c:9040-9115) */
static float dot_q6_K_q8_K(int64_t n, const void *vx, const void *vy) {
    const blk_q6_K *x = vx; const blk_q8_K *y = vy; float acc[8] = {0};
    for (int64_t i = 0; i < n / QK_K; ++i) {
        const float d = y[i].d * pso_fp16_to_fp32(x[i].d);
        const uint8_t *ql = x[i].ql, *qh = x[i].qh; const int8_t *q8 = y[i].qs, *sc = x[i].scales;
        int sumi[8] = {0}; int is = 0;
        for (int j = 0; j < QK_K / 128; ++j) {
            for (int sub = 0; sub < 4; sub++) { /* q4_0..q4_3: elements sub*32 .. sub*32+31 of this 128-chunk */
                for (int u = 0; u < 8; u++) {
                    int s = 0;
                    for (int e = 4 * u; e < 4 * u + 4; e++) {
                        int lo = (sub & 1) ? ql[32 + e] : ql[e];
                        lo = (sub >= 2) ? (lo >> 4) : (lo & 0xF);
                        int q = (lo | (((qh[e] >> (2 * sub)) & 3) << 4)) - 32;
                        s += q * q8[sub * 32 + e];
                    }
                    sumi[u] += sc[2 * (is + sub) + (u >= 4)] * s;
                }
            }
            is += 4; ql += 64; qh += 32; q8 += 128;
        }
        for (int u = 0; u < 8; u++) acc[u] = fmaf(d, (float)sumi[u], acc[u]);
    }
    return hsum8(acc);
}

/* ggml_vec_dot_f32 with the AVX macros (ggml.c:2092-2133, :1335-1383): 4 accumulators x 8 lanes, FMA,
 * GGML_F32x8_REDUCE association, scalar leftovers. */
float pso_vec_dot_f32(int64_t n, const float *x, const float *y) {
    float sum[4][8]; memset(sum, 0, sizeof(sum));
    const int64_t np = n & ~(int64_t)31;
    for (int64_t i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 8; l++) sum[j][l] = fmaf(x[i + j * 8 + l], y[i + j * 8 + l], sum[j][l]);
    for (int l = 0; l < 8; l++) { sum[0][l] += sum[2][l]; sum[1][l] += sum[3][l]; }
    for (int l = 0; l < 8; l++) sum[0][l] += sum[1][l];
    float t0[4]; for (int l = 0; l < 4; l++) t0[l] = sum[0][l] + sum[0][l + 4];
    float sumf = (t0[0] + t0[1]) + (t0[2] + t0[3]);
    /* contracted build: GCC vectorises this loop (8 products at a time, then 4: vmulps + in-order vaddss, NOT fused) and fuses only the
     * last (n - np) % 4 scalar steps (vfmadd231ss) */
    const int64_t nf = g_contract ? np + ((n - np) & ~(int64_t)3) : n;
    for (int64_t i = np; i < nf; ++i) sumf += x[i] * y[i];
    for (int64_t i = nf; i < n; ++i) sumf = fmaf(x[i], y[i], sumf);
    return sumf;
}

float pso_vec_dot(int type, int64_t n, const void *vx, const void *vy) {
    switch (type) {
    case PSO_Q4_0: return dot_q4_0_q8_0(n, vx, vy);
    case PSO_Q8_0: return dot_q8_0_q8_0(n, vx, vy);
    case PSO_Q4_K: return dot_q4_K_q8_K(n, vx, vy);
    case PSO_Q5_K: return dot_q5_K_q8_K(n, vx, vy);
    case PSO_Q6_K: return dot_q6_K_q8_K(n, vx, vy);
    case PSO_F32: return pso_vec_dot_f32(n, vx, vy);
    }
    return NAN;
}

/* ------------------------------------------------------------------ tiny parallel-for */
typedef struct { void (*fn)(void *, int64_t, int64_t); void *arg; int64_t lo, hi; } pf_task;
static void *pf_tramp(void *p) { pf_task *t = p; t->fn(t->arg, t->lo, t->hi); return NULL; }
static void parallel_for(int nth, int64_t n, void (*fn)(void *, int64_t, int64_t), void *arg) {
    if (nth <= 1 || n < 2 * nth) { fn(arg, 0, n); return; }
    pthread_t th[64]; pf_task t[64]; if (nth > 64) nth = 64;
    for (int i = 0; i < nth; i++) {
        t[i].fn = fn; t[i].arg = arg; t[i].lo = n * i / nth; t[i].hi = n * (i + 1) / nth;
        pthread_create(&th[i], NULL, pf_tramp, &t[i]);
    }
    for (int i = 0; i < nth; i++) pthread_join(th[i], NULL);
}

/* ------------------------------------------------------------------ ops */
typedef struct { int type; const char *w; size_t wrs; int64_t K, N, bs; const char *act; size_t ars; float *y; } mm_args;
static void mm_rows(void *p, int64_t lo, int64_t hi) {
    mm_args *a = p;
    for (int64_t r = lo; r < hi; r++)
        for (int64_t c = 0; c < a->bs; c++)
            a->y[c * a->N + r] = pso_vec_dot(a->type, a->K, a->w + r * a->wrs, a->act + c * a->ars);
}
/* powerserve_compute_forward_mul_mat (ggml.c:13434-13648): quantize every activation row to the
 * vec_dot_type (:13502-13530), then one vec_dot per (row, col) (:13391-13431).  Each output element is
 * produced by exactly one vec_dot, so the result is independent of the thread partition. */
void pso_mul_mat(int type, const void *w, int64_t K, int64_t N, const float *x, int64_t bs, float *y, void *act_out,
                 int n_threads) {
    const int vdt = pso_vec_dot_type(type); const size_t ars = pso_row_size(vdt, K);
    char *act = act_out ? (char *)act_out : malloc(ars * bs);
    for (int64_t c = 0; c < bs; c++) pso_from_float(vdt, x + c * K, act + c * ars, K);
    mm_args a = {type, w, pso_row_size(type, K), K, N, bs, act, ars, y};
    parallel_for(n_threads, N, mm_rows, &a);
    if (!act_out) free(act);
}

/* powerserve_compute_forward_rms_norm_f32 (ggml.c:12667-12720) + ggml_vec_scale_f32_weight (:2442-2470) */
void pso_rms_norm(const float *x, const float *w, float *y, int64_t ne0, int64_t nrows, float eps) {
    for (int64_t r = 0; r < nrows; r++) {
        const float *xr = x + r * ne0; float *yr = y + r * ne0;
        double sum = 0.0;
        for (int64_t i = 0; i < ne0; i++) sum += (double)(xr[i] * xr[i]);
        const float mean = (float)(sum / ne0);
        const float scale = 1.0f / sqrtf(mean + eps);
        for (int64_t i = 0; i < ne0; i++) yr[i] = xr[i] * (w[i] * scale);
    }
}

/* rope: ggml_rope_cache_init (ggml.c:15344-15358), rope_yarn (:15319-15336), corr dims (:15360-15366),
 * ggml_compute_forward_rope_f32 (:15368-15491); freq_factors == NULL always (ggml_wrapper.cpp:104-106). */
static float rope_yarn_ramp(const float low, const float high, const int i0) {
    const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low);
    return 1 - fminf(1, fmaxf(0, y));
}
static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float)M_PI)) / (2 * logf(base));
}
void pso_rope_cache(int32_t p, int64_t ne0, const pso_rope_params *rp, float *cache) {
    const float theta_scale = powf(rp->freq_base, -2.0f / rp->n_dims);
    float corr[2];
    float start = floorf(rope_corr_dim(rp->n_dims, rp->n_ctx_orig, rp->beta_fast, rp->freq_base));
    float end   = ceilf(rope_corr_dim(rp->n_dims, rp->n_ctx_orig, rp->beta_slow, rp->freq_base));
    corr[0] = fmaxf(0, start); corr[1] = fminf(rp->n_dims - 1, end);
    float theta = (float)p;
    for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
        float theta_extrap = theta, mscale = rp->attn_factor;
        float theta_interp = rp->freq_scale * theta_extrap, th = theta_interp;
        if (rp->ext_factor != 0.0f) {
            float ramp_mix = rope_yarn_ramp(corr[0], corr[1], (int)i0) * rp->ext_factor;
            th = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
            mscale *= 1.0f + 0.1f * logf(1.0f / rp->freq_scale);
        }
        cache[i0 + 0] = cosf(th) * mscale;
        cache[i0 + 1] = sinf(th) * mscale;
        cache[i0 + 1] *= 1.0f; /* sin_sign, forward */
        theta *= theta_scale;
    }
}
/* x0 * c - x1 * sn and x0 * sn + x1 * c as the contracted build evaluates them: GCC fuses the FIRST product of each expression with the
 * add / subtract and leaves the second one a rounded multiply (vfmsub / vfmadd; the other three pairings were tried against libps_ref_fast.so and
 * differ in 15-31 % of the outputs) */
#define ROPE_CONTRACTED(out0, out1) do { (out0) = fmaf(x0, c, -(x1 * sn)); (out1) = fmaf(x0, sn, x1 * c); } while (0)
void pso_rope(const float *src, float *dst, int64_t ne0, int64_t ne1, int64_t ne2, const int32_t *pos,
              const pso_rope_params *rp) {
    float *cache = malloc(sizeof(float) * (size_t)ne0);
    const int n_dims = rp->n_dims; const int is_neox = rp->mode & 2;
    for (int64_t i2 = 0; i2 < ne2; i2++) {
        pso_rope_cache(pos[i2], ne0, rp, cache);
        for (int64_t i1 = 0; i1 < ne1; i1++) {
            const float *s = src + (i2 * ne1 + i1) * ne0; float *d = dst + (i2 * ne1 + i1) * ne0;
            if (!is_neox) {
                for (int64_t i0 = 0; i0 < n_dims; i0 += 2) {
                    const float c = cache[i0], sn = cache[i0 + 1], x0 = s[i0], x1 = s[i0 + 1];
                    if (g_contract) { ROPE_CONTRACTED(d[i0], d[i0 + 1]); continue; }
                    d[i0] = x0 * c - x1 * sn; d[i0 + 1] = x0 * sn + x1 * c;
                }
            } else {
                for (int64_t i0 = 0; i0 < n_dims; i0 += 2) {
                    const int64_t ic = i0 / 2; const float c = cache[i0], sn = cache[i0 + 1];
                    const float x0 = s[ic], x1 = s[ic + n_dims / 2];
                    if (g_contract) { ROPE_CONTRACTED(d[ic], d[ic + n_dims / 2]); continue; }
                    d[ic] = x0 * c - x1 * sn; d[ic + n_dims / 2] = x0 * sn + x1 * c;
                }
            }
            for (int64_t i0 = n_dims; i0 < ne0; i0++) d[i0] = s[i0];
        }
    }
    free(cache);
}

/* ggml_v_expf, AVX2+FMA variant (ggml.c:2685-2723), one lane */
static float v_expf1(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    uint32_t zb; memcpy(&zb, &z, 4);
    const uint32_t e = zb << 23;
    uint32_t kb = e + 0x3f800000u; float k; memcpy(&k, &kb, 4);
    const int c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u,
                         0x1.ffffecp-1f * b);
    if (!c) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    uint32_t s1b = g + 0x7f000000u, s2b = e - g; float s1, s2; memcpy(&s1, &s1b, 4); memcpy(&s2, &s2b, 4);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}

/* powerserve_compute_forward_softmax_ext (ggml.c:15091-15117) -> ggml_compute_forward_soft_max_f32
 * (:14846-14940) with max_bias = 0 (slope 1), ggml_vec_soft_max_f32 (:2814-2863). */
void pso_softmax_ext(const float *x, const float *mask, float *out, int64_t n_kv, int64_t bs, int64_t n_heads,
                     float scale) {
    float *wp = malloc(sizeof(float) * (size_t)n_kv);
    for (int64_t r = 0; r < bs * n_heads; r++) {
        const float *sp = x + r * n_kv; float *dp = out + r * n_kv; const float *mp = mask ? mask + (r % bs) * n_kv : NULL;
        for (int64_t i = 0; i < n_kv; i++) wp[i] = sp[i] * scale;
        if (mp) for (int64_t i = 0; i < n_kv; i++) wp[i] += 1.0f * mp[i];
        float max = -INFINITY; for (int64_t i = 0; i < n_kv; i++) max = fmaxf(max, wp[i]);
        double sum = 0; int64_t i = 0;
        for (; i + 7 < n_kv; i += 8) {
            float v[8]; for (int l = 0; l < 8; l++) { v[l] = v_expf1(wp[i + l] - max); dp[i + l] = v[l]; }
            float a0 = v[4] + v[0], a1 = v[5] + v[1], a2 = v[6] + v[2], a3 = v[7] + v[3];
            a0 = a0 + a2; a1 = a1 + a3; sum += (double)(a0 + a1);
        }
        for (; i < n_kv; ++i) { float val = expf(wp[i] - max); sum += (double)val; dp[i] = val; }
        sum = 1.0 / sum; const float inv = (float)sum;
        for (int64_t q = 0; q < n_kv; q++) dp[q] *= inv;
    }
    free(wp);
}

/* GGMLBackend::silu_hadamard (src/backend/ggml/ggml.cpp:115-129) */
void pso_silu_hadamard(const float *gate, const float *up, float *out, int64_t n) {
    for (int64_t j = 0; j < n; j++) {
        float val = gate[j];
        val *= (1.0f / (1.0f + expf(-val)));
        val *= up[j];
        out[j] = val;
    }
}

/* powerserve_compute_forward_add_f32 (ggml.c:10042-10115): row-broadcast of b when it has one row */
void pso_add(const float *a, const float *b, float *out, int64_t ne0, int64_t nrows, int b_is_row) {
    for (int64_t r = 0; r < nrows; r++)
        for (int64_t i = 0; i < ne0; i++) out[r * ne0 + i] = a[r * ne0 + i] + b[(b_is_row ? 0 : r * ne0) + i];
}

/* GGMLBackend::get_embedding (src/backend/ggml/ggml_wrapper.cpp:181-211); the reference supports
 * F32/Q4_0/Q8_0 and aborts otherwise — K-quant tables are an extension of the new backend, dequantized
 * with the matching dequantize_row_* . */
void pso_get_embedding(int type, const void *table, int64_t dim, const int32_t *tokens, int n, float *out) {
    const size_t rs = pso_row_size(type, dim);
    for (int i = 0; i < n; i++) pso_dequantize_row(type, (const char *)table + rs * (size_t)tokens[i], out + (size_t)i * dim, dim);
}

/* ------------------------------------------------------------------ whole model */
typedef struct { int type; const void *data; int64_t ne0, ne1; } pso_w;
typedef struct {
    pso_w attn_norm, ffn_norm, attn_q, attn_k, attn_v, attn_output, ffn_gate, ffn_up, ffn_down;
    pso_w attn_q_bias, attn_k_bias, attn_v_bias;
} pso_layer;
struct pso_model {
    pso_llm_config cfg; int is_qwen2, n_threads;
    pso_w token_embd, output, output_norm; pso_layer *lw;
    float **k_cache, **v_cache; /* per layer: K [n_ctx][kv_dim]; V [kv_dim][n_ctx] (ggml_kv_cache.cpp:43-57,
                                   norm_attention.cpp:82-104) */
    size_t position;
};

pso_model *pso_model_create(const pso_llm_config *cfg, int is_qwen2, int n_threads) {
    pso_model *m = calloc(1, sizeof(*m));
    m->cfg = *cfg; m->is_qwen2 = is_qwen2; m->n_threads = n_threads;
    m->lw = calloc(cfg->n_layers, sizeof(pso_layer));
    m->k_cache = calloc(cfg->n_layers, sizeof(float *)); m->v_cache = calloc(cfg->n_layers, sizeof(float *));
    for (uint32_t L = 0; L < cfg->n_layers; L++) {
        m->k_cache[L] = calloc((size_t)cfg->seq_len * cfg->kv_dim, sizeof(float));
        m->v_cache[L] = calloc((size_t)cfg->seq_len * cfg->kv_dim, sizeof(float));
    }
    return m;
}
void pso_model_destroy(pso_model *m) {
    for (uint32_t L = 0; L < m->cfg.n_layers; L++) { free(m->k_cache[L]); free(m->v_cache[L]); }
    free(m->k_cache); free(m->v_cache); free(m->lw); free(m);
}
/* tensor names: src/model/common/weights.hpp:26-69, llama_weight.hpp:25-33, qwen2_weight.hpp:25-36 */
int pso_model_set_tensor(pso_model *m, const char *name, int type, const void *data, int64_t ne0, int64_t ne1) {
    pso_w w = {type, data, ne0, ne1};
    if (!strcmp(name, "token_embd.weight")) { m->token_embd = w; return 0; }
    if (!strcmp(name, "output.weight")) { m->output = w; return 0; }
    if (!strcmp(name, "output_norm.weight")) { m->output_norm = w; return 0; }
    int L; char rest[64];
    if (sscanf(name, "blk.%d.%63s", &L, rest) == 2 && L >= 0 && (uint32_t)L < m->cfg.n_layers) {
        pso_layer *l = &m->lw[L];
#define SETW(nm, field) if (!strcmp(rest, nm)) { l->field = w; return 0; }
        SETW("attn_norm.weight", attn_norm) SETW("ffn_norm.weight", ffn_norm) SETW("attn_q.weight", attn_q)
        SETW("attn_k.weight", attn_k) SETW("attn_v.weight", attn_v) SETW("attn_output.weight", attn_output)
        SETW("ffn_gate.weight", ffn_gate) SETW("ffn_up.weight", ffn_up) SETW("ffn_down.weight", ffn_down)
        SETW("attn_q.bias", attn_q_bias) SETW("attn_k.bias", attn_k_bias) SETW("attn_v.bias", attn_v_bias)
#undef SETW
    }
    return -1; /* unknown tensors (e.g. rope_freqs.weight) are ignored by the reference too (§0.6) */
}
size_t pso_model_kv_position(const pso_model *m) { return m->position; }
void pso_model_reset(pso_model *m) { m->position = 0; } /* truncate_tokens(kv_size = 0) */
void pso_model_rollback(pso_model *m, size_t n) { m->position -= n < m->position ? n : m->position; } /* rollback_tokens (kv_cache.hpp:256-264) */
const float *pso_model_k_cache(const pso_model *m, int L) { return m->k_cache[L]; }
const float *pso_model_v_cache(const pso_model *m, int L) { return m->v_cache[L]; }

static void mm(pso_model *m, const pso_w *w, const float *x, int64_t bs, float *y) {
    pso_mul_mat(w->type, w->data, w->ne0, w->ne1, x, bs, y, NULL, m->n_threads);
}

typedef struct {
    pso_model *m; int L; int64_t bs, n_kv; const float *q; const float *mask /* [bs][n_kv]: 0 or -inf */; float *att_out;
} attn_args;
/* per q-head: KQ (F32 mat-mul of the K-cache view, norm_attention.cpp:115-129), softmax_ext with the caller's mask
 * (norm_attention.cpp:130-134; ggml.c:14901-14914 honours any mask), V·kq (:138-147), permute+cont (:149-151) */
static void attn_heads(void *p, int64_t lo, int64_t hi) {
    attn_args *a = p; const pso_llm_config *c = &a->m->cfg;
    const int64_t hs = c->head_size, kvd = c->kv_dim, nctx = c->seq_len, dim = (int64_t)c->n_heads * hs;
    const int64_t r2 = c->n_heads / c->n_kv_heads, bs = a->bs, n_kv = a->n_kv;
    const float scale = 1.0f / sqrtf((float)hs);
    float *kq = malloc(sizeof(float) * (size_t)(n_kv * bs)), *sm = malloc(sizeof(float) * (size_t)(n_kv * bs));
    const float *mask = a->mask;
    for (int64_t h = lo; h < hi; h++) {
        const int64_t kvh = h / r2;
        const float *K = a->m->k_cache[a->L]; const float *V = a->m->v_cache[a->L];
        for (int64_t i = 0; i < bs; i++)
            for (int64_t j = 0; j < n_kv; j++)
                kq[i * n_kv + j] = pso_vec_dot_f32(hs, K + j * kvd + kvh * hs, a->q + i * dim + h * hs);
        pso_softmax_ext(kq, mask, sm, n_kv, bs, 1, scale);
        for (int64_t i = 0; i < bs; i++)
            for (int64_t d = 0; d < hs; d++)
                a->att_out[i * dim + h * hs + d] = pso_vec_dot_f32(n_kv, V + (kvh * hs + d) * nctx, sm + i * n_kv);
    }
    free(kq); free(sm);
}

/* LlamaModel::forward (src/model/llama/llama_model.cpp:52-117) with NormAttention::build
 * (src/model/module/norm_attention.cpp:26-160) and FFN::build (src/model/module/ffn.cpp:22-42) */
/* the forward with everything the graph derives from `pos` made explicit: cache slots [slot0, slot0 + n), RoPE positions,
 * n_kv and the additive mask [n][n_kv] */
static int forward_impl(pso_model *m, const int32_t *tokens, int n, size_t slot0, const int32_t *pos, int64_t n_kv, const float *mask,
                        int lm_head, float *logits_out, int advance) {
    const pso_llm_config *c = &m->cfg; const int64_t bs = n, dim = c->dim, kvd = c->kv_dim, hid = c->hidden_dim;
    const int64_t nctx = c->seq_len, hs = c->head_size;
    if (slot0 + (size_t)n > (size_t)nctx) return -1;
    float *x = malloc(sizeof(float) * bs * dim), *nrm = malloc(sizeof(float) * bs * dim);
    float *q = malloc(sizeof(float) * bs * dim), *k = malloc(sizeof(float) * bs * kvd), *v = malloc(sizeof(float) * bs * kvd);
    float *qr = malloc(sizeof(float) * bs * dim), *kr = malloc(sizeof(float) * bs * kvd);
    float *att = malloc(sizeof(float) * bs * dim), *ao = malloc(sizeof(float) * bs * dim);
    float *g = malloc(sizeof(float) * bs * hid), *u = malloc(sizeof(float) * bs * hid), *hb = malloc(sizeof(float) * bs * hid);
    float *dn = malloc(sizeof(float) * bs * dim);
    pso_get_embedding(m->token_embd.type, m->token_embd.data, dim, tokens, n, x);
    const size_t cur_pos = slot0;
    for (uint32_t L = 0; L < c->n_layers; L++) {
        pso_layer *l = &m->lw[L];
        pso_rms_norm(x, l->attn_norm.data, nrm, dim, bs, c->norm_eps);
        mm(m, &l->attn_q, nrm, bs, q); mm(m, &l->attn_k, nrm, bs, k); mm(m, &l->attn_v, nrm, bs, v);
        if (m->is_qwen2) {
            pso_add(q, l->attn_q_bias.data, q, dim, bs, 1); pso_add(k, l->attn_k_bias.data, k, kvd, bs, 1);
            pso_add(v, l->attn_v_bias.data, v, kvd, bs, 1);
        }
        pso_rope(q, qr, hs, c->n_heads, bs, pos, &c->rope);
        pso_rope(k, kr, hs, c->n_kv_heads, bs, pos, &c->rope);
        /* store kv: K rows [cur_pos, cur_pos+bs); V transposed scatter at column cur_pos (norm_attention.cpp:78-105) */
        memcpy(m->k_cache[L] + cur_pos * kvd, kr, sizeof(float) * bs * kvd);
        for (int64_t i = 0; i < bs; i++)
            for (int64_t d = 0; d < kvd; d++) m->v_cache[L][d * nctx + cur_pos + i] = v[i * kvd + d];
        attn_args a = {m, (int)L, bs, n_kv, qr, mask, att};
        parallel_for(m->n_threads, c->n_heads, attn_heads, &a);
        mm(m, &l->attn_output, att, bs, ao);
        pso_add(x, ao, x, dim, bs, 0);
        pso_rms_norm(x, l->ffn_norm.data, nrm, dim, bs, c->norm_eps);
        mm(m, &l->ffn_gate, nrm, bs, g); mm(m, &l->ffn_up, nrm, bs, u);
        pso_silu_hadamard(g, u, hb, bs * hid);
        mm(m, &l->ffn_down, hb, bs, dn);
        pso_add(x, dn, x, dim, bs, 0);
    }
    if (lm_head) {
        pso_rms_norm(x, m->output_norm.data, nrm, dim, bs, c->norm_eps);
        const pso_w *ow = m->output.data ? &m->output : &m->token_embd; /* tied lm_head (weights.hpp:67-68) */
        mm(m, ow, nrm, bs, logits_out);
    }
    if (advance) m->position += (size_t)n; /* m_kv->advance(batch_size) (llama_model.cpp:109) */
    free(x); free(nrm); free(q); free(k); free(v); free(qr); free(kr); free(att); free(ao); free(g); free(u); free(hb); free(dn);
    return 0;
}

/* LlamaModel::forward as the reference runs it: slots = positions, n_kv = pos.back() + 1 (norm_attention.cpp:111), mask
 * j <= pos[i] (executor.cpp:210-224) */
int pso_model_forward(pso_model *m, const int32_t *tokens, int n, const int32_t *pos, int lm_head, float *logits_out) {
    const int64_t n_kv = pos[n - 1] + 1;
    float *mask = malloc(sizeof(float) * (size_t)(n_kv * n));
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < n_kv; j++) mask[i * n_kv + j] = (j <= pos[i]) ? 0.f : -INFINITY;
    const int rc = forward_impl(m, tokens, n, (size_t)pos[0], pos, n_kv, mask, lm_head, logits_out, 1);
    free(mask);
    return rc;
}

/* Token-tree forward (src/speculative/token_tree.cpp:181-234 verify, :297-315 the tree mask; the QNN backend's mask fill,
 * src/backend/qnn/causal_models.hpp:117-128): the n tokens go to the cache slots [position, position + n), column i is rotated
 * with rope_pos[i] (its depth in the tree, not its slot), sees the cached slot j < position iff kv_vis[j] (KVCacheInterface::mask;
 * NULL: all) and the batch column j iff tree[i * n + j] (NULL: causal).  The same op sequence as pso_model_forward — softmax_ext
 * takes any mask (ggml.c:14901-14914); only the CPU executor's GET_MASK cannot express this one. */
int pso_model_forward_tree(pso_model *m, const int32_t *tokens, int n, const int32_t *rope_pos, const uint8_t *tree,
                           const uint8_t *kv_vis, int lm_head, float *logits_out, int advance) {
    const int64_t p0 = (int64_t)m->position, n_kv = p0 + n;
    float *mask = malloc(sizeof(float) * (size_t)(n_kv * n));
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < n_kv; j++) {
            const int vis = j < p0 ? (kv_vis ? kv_vis[j] != 0 : 1) : (tree ? tree[i * n + (j - p0)] != 0 : (j - p0) <= i);
            mask[i * n_kv + j] = vis ? 0.f : -INFINITY;
        }
    const int rc = forward_impl(m, tokens, n, (size_t)p0, rope_pos, n_kv, mask, lm_head, logits_out, advance);
    free(mask);
    return rc;
}
void pso_model_kv_advance(pso_model *m, size_t n) { m->position += n; } /* advance_tokens (kv_cache.hpp:249-255) */
/* KVCacheInterface::move (kv_cache.hpp): slot src -> slot dst in every layer */
void pso_model_kv_move(pso_model *m, size_t dst, size_t src) {
    const pso_llm_config *c = &m->cfg;
    for (uint32_t L = 0; L < c->n_layers; L++) {
        memcpy(m->k_cache[L] + dst * c->kv_dim, m->k_cache[L] + src * c->kv_dim, sizeof(float) * c->kv_dim);
        for (uint32_t d = 0; d < c->kv_dim; d++) m->v_cache[L][(size_t)d * c->seq_len + dst] = m->v_cache[L][(size_t)d * c->seq_len + src];
    }
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* ModelTokenIterator (src/model/model.hpp:117-184) + greedy arg-max (prob_array.cpp:65-67) */
int pso_model_generate(pso_model *m, const int32_t *prompt, int n_prompt, int batch_size, int steps,
                       int32_t *out_tokens, float *logits_out, double *t_prefill_s, double *t_decode_s) {
    pso_model_reset(m);
    const size_t V = m->cfg.vocab_size; float *lg = malloc(sizeof(float) * V);
    double t0 = now_s(); int n_prefilled = 0;
    while (n_prefilled < n_prompt - 1) {
        int bs = batch_size < n_prompt - n_prefilled - 1 ? batch_size : n_prompt - n_prefilled - 1;
        int32_t *pos = malloc(sizeof(int32_t) * bs);
        for (int i = 0; i < bs; i++) pos[i] = (int32_t)m->position + i;
        if (pso_model_forward(m, prompt + n_prefilled, bs, pos, 0, NULL)) { free(pos); free(lg); return -1; }
        free(pos); n_prefilled += bs;
    }
    double t1 = now_s(); int32_t cur = prompt[n_prompt - 1];
    for (int s = 0; s < steps; s++) {
        int32_t pos = (int32_t)m->position;
        if (pso_model_forward(m, &cur, 1, &pos, 1, lg)) { free(lg); return -1; }
        size_t best = 0; for (size_t i = 1; i < V; i++) if (lg[i] > lg[best]) best = i;
        if (logits_out) memcpy(logits_out + (size_t)s * V, lg, sizeof(float) * V);
        out_tokens[s] = (int32_t)best; cur = (int32_t)best;
    }
    double t2 = now_s();
    if (t_prefill_s) *t_prefill_s = t1 - t0;
    if (t_decode_s) *t_decode_s = t2 - t1;
    free(lg); return 0;
}
