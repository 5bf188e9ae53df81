// Device code shared by the quantized mat-vec kernels (k_gemv.hip: gemv / gemv1 / gemv3 / gemm8*, k_gemv4.hip: gemv4):
// weight-type traits, the LDS image of an activation column, the per-unit integer work (unit_dot, unit_rec), the fp32
// chain step (rec_chain) and hsum_float_8 (row_reduce).  See k_gemv.hip's header for the numerics contract.
#pragma once
#include <type_traits>
#include "ps_dev.h"
#include "ps_internal.h"
#include "ps_ops.h"
#include "ps_quant_dev.h"

namespace {

template <int WT> struct WTraits;
template <> struct WTraits<PS_Q4_0> { static constexpr int RG = 16, UNIT = 128, BLK = 32,  VDT = PS_Q8_0; };
template <> struct WTraits<PS_Q8_0> { static constexpr int RG = 8,  UNIT = 128, BLK = 32,  VDT = PS_Q8_0; };
template <> struct WTraits<PS_Q4_K> { static constexpr int RG = 8,  UNIT = 256, BLK = 256, VDT = PS_Q8_K; };

struct LAct { // LDS image of one activation column (natural element order)
    const int *q32;   // int8 quants viewed as dwords
    const float *d;   // per-block scale
    const int *bs32;  // sums of 32 consecutive quants
};

__device__ __forceinline__ int bfe8(uint32_t v, int byte) { return (int)((v >> (8 * byte)) & 0xff); }

// one unit (1 KiB of 8/16 rows) against one activation column; acc0/acc1/accm are this lane's fma chains
template <int WT>
__device__ __forceinline__ void unit_dot(const uint4 q, const uint4 h, const int unit, const int u, const LAct a,
                                         float &acc0, float &acc1, float &accm) {
    constexpr uint32_t M = 0x0F0F0F0Fu;
    if (WT == PS_Q4_K) {
        // h = {d|dmin, scales[0..3], scales[4..7], scales[8..11]}; 6-bit unpack as ggml-quants.c:7818-7823
        const uint32_t sc03 = h.y & 0x3f3f3f3fu;
        const uint32_t sc47 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
        const uint32_t mn03 = h.z & 0x3f3f3f3fu;
        const uint32_t mn47 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
        const int base = unit * 64 + u; // dword index of element unit*256 + u*4
        const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
        int s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int yl = a.q32[base + j * 16], yh = a.q32[base + j * 16 + 8];
            const uint32_t scp = (j < 2) ? sc03 : sc47;
            const int sl = bfe8(scp, (2 * j) & 3), sh = bfe8(scp, (2 * j + 1) & 3);
            s += sl * dot4((int)(wq[j] & M), yl, 0) + sh * dot4((int)((wq[j] >> 4) & M), yh, 0);
        }
        const float yd   = a.d[unit];
        const float d    = __fmul_rn(yd, ps_h2f((uint16_t)(h.x & 0xffff)));
        const float dmin = __fmul_rn(-yd, ps_h2f((uint16_t)(h.x >> 16)));
        acc0 = __fmaf_rn(d, (float)s, acc0);
        // acc_m lane v = u & 3: prod = mins[2v]*q8s[2v] + mins[2v+1]*q8s[2v+1]   (ggml-quants.c:7831-7834)
        const int v = u & 3;
        const uint32_t mp = (v < 2) ? mn03 : mn47;
        const int prod = bfe8(mp, (2 * v) & 3) * a.bs32[unit * 8 + 2 * v] + bfe8(mp, (2 * v + 1) & 3) * a.bs32[unit * 8 + 2 * v + 1];
        accm = __fmaf_rn(dmin, (float)prod, accm);
    } else if (WT == PS_Q8_0) {
        // q = quad u of blocks 4*unit .. 4*unit+3; h.x,h.y = their four fp16 scales
        const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
        const uint16_t dh[4] = {(uint16_t)(h.x & 0xffff), (uint16_t)(h.x >> 16), (uint16_t)(h.y & 0xffff), (uint16_t)(h.y >> 16)};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int blk = unit * 4 + b;
            const int s = dot4((int)wq[b], a.q32[blk * 8 + u], 0);
            acc0 = __fmaf_rn(__fmul_rn(ps_h2f(dh[b]), a.d[blk]), (float)s, acc0);
        }
    } else { // Q4_0: lane u' (0..3) holds bytes 4u'..4u'+3 of each block: low nibbles = quad u', high = quad u'+4
        const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
        const uint16_t dh[4] = {(uint16_t)(h.x & 0xffff), (uint16_t)(h.x >> 16), (uint16_t)(h.y & 0xffff), (uint16_t)(h.y >> 16)};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int blk = unit * 4 + b;
            const int yl = a.q32[blk * 8 + u], yh = a.q32[blk * 8 + 4 + u];
            const int sl = dot4((int)(wq[b] & M), yl, 0) - 8 * dot4(0x01010101, yl, 0);        // sum (q-8)*y
            const int sh = dot4((int)((wq[b] >> 4) & M), yh, 0) - 8 * dot4(0x01010101, yh, 0);
            const float d = __fmul_rn(ps_h2f(dh[b]), a.d[blk]);
            acc0 = __fmaf_rn(d, (float)sl, acc0);
            acc1 = __fmaf_rn(d, (float)sh, acc1);
        }
    }
}

// hsum_float_8 association (+ acc_m for Q4_K): valid in the lane with u == 0
template <int WT>
__device__ __forceinline__ float row_reduce(float acc0, float acc1, float accm) {
    // only u == 0 keeps the result: one-directional DPP row shifts (lane u reads lane u + n inside its row of 16) give the
    // same partners and association as the xor butterfly without any ds_bpermute round trip on the chain waves' path
    if (WT == PS_Q4_0) {
        float r = __fadd_rn(acc1, acc0); // a[k+4] + a[k]
        r = __fadd_rn(r, dpp_f<0x102>(r));
        r = __fadd_rn(r, dpp_f<0x101>(r));
        return r;
    }
    float r = __fadd_rn(acc0, dpp_f<0x104>(acc0));
    r = __fadd_rn(r, dpp_f<0x102>(r));
    r = __fadd_rn(r, dpp_f<0x101>(r));
    if (WT == PS_Q4_K) {
        float m = __fadd_rn(accm, dpp_f<0x102>(accm)); // (m0+m2), (m1+m3)
        m = __fadd_rn(m, dpp_f<0x101>(m));
        r = __fadd_rn(r, m);
    }
    return r;
}


template <int WT> struct RecOf { using T = int4; };
template <> struct RecOf<PS_Q4_K> { using T = int2; };

// Q4_K record: {s, u < 4 ? prod : d|dmin}: the four acc_m lanes need prod, the other four carry the fp16 pair
// QT (Q4_K, k_gemv4.hip): the quants of a 256-element tile are stored quad-major in LDS — dword u * 8 + g instead of
// g * 8 + u (g = sub-block, u = AVX lane) — so that lane u fetches its eight dwords with two ds_read_b128 instead of
// eight ds_read_b32
template <int WT, bool QT = false>
__device__ __forceinline__ typename RecOf<WT>::T unit_rec(const uint4 q, const uint4 h, const int unit, const int u, const LAct a) {
    constexpr uint32_t M = 0x0F0F0F0Fu;
    const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
    if constexpr (WT == PS_Q4_K) {
        const uint32_t sc03 = h.y & 0x3f3f3f3fu;
        const uint32_t sc47 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
        const uint32_t mn03 = h.z & 0x3f3f3f3fu;
        const uint32_t mn47 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
        const int base = unit * 64 + u;
        int s = 0;
        int ylv[4], yhv[4]; // every LDS read of the unit first: one wait instead of one per 64-element group
        const int v = u & 3;
        int bsa, bsb;
        if constexpr (QT) {
            const int4 y0 = *(const int4 *)(a.q32 + unit * 64 + u * 8), y1 = *(const int4 *)(a.q32 + unit * 64 + u * 8 + 4);
            ylv[0] = y0.x; yhv[0] = y0.y; ylv[1] = y0.z; yhv[1] = y0.w; ylv[2] = y1.x; yhv[2] = y1.y; ylv[3] = y1.z; yhv[3] = y1.w;
            const int2 bs = *(const int2 *)(a.bs32 + unit * 8 + 2 * v);
            bsa = bs.x; bsb = bs.y;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) { ylv[j] = a.q32[base + j * 16]; yhv[j] = a.q32[base + j * 16 + 8]; }
            bsa = a.bs32[unit * 8 + 2 * v]; bsb = a.bs32[unit * 8 + 2 * v + 1];
        }
        int dlo[4], dhi[4]; // the eight quad dots as plain v_dot4 (no zeroed accumulators)
        dot4x4(dlo, (int)(wq[0] & M), (int)(wq[1] & M), (int)(wq[2] & M), (int)(wq[3] & M), ylv[0], ylv[1], ylv[2], ylv[3]);
        dot4x4(dhi, (int)((wq[0] >> 4) & M), (int)((wq[1] >> 4) & M), (int)((wq[2] >> 4) & M), (int)((wq[3] >> 4) & M), yhv[0], yhv[1], yhv[2], yhv[3]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // |dot4| <= 4*15*127 fits int16, the 6-bit scales too: the two quad dots of a 64-element group are packed into
            // one dword and meet their scale pair in a single v_dot2_i32_i16 (exact integer arithmetic, 32-bit accumulate)
            const uint32_t scp  = (j < 2) ? sc03 : sc47;
            const uint32_t sc16 = __builtin_amdgcn_perm(0u, scp, (j & 1) ? 0x0c030c02u : 0x0c010c00u); // {scale[2j], scale[2j+1]} as int16
            const uint32_t d16 = __builtin_amdgcn_perm((uint32_t)dhi[j], (uint32_t)dlo[j], 0x05040100u); // {dl, dh} as int16
            s = dot2_i16(d16, sc16, s);
        }
        const uint32_t mp = (v < 2) ? mn03 : mn47;
        const int pr = __mul24(bfe8(mp, (2 * v) & 3), bsa) + __mul24(bfe8(mp, (2 * v + 1) & 3), bsb);
        return make_int2(s, u < 4 ? pr : (int)h.x);
    } else if constexpr (WT == PS_Q8_0) {
        int s[4], y[4];
#pragma unroll
        for (int b = 0; b < 4; b++) y[b] = a.q32[(unit * 4 + b) * 8 + u];
        dot4x4(s, (int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3], y[0], y[1], y[2], y[3]);
        return make_int4(s[0], s[1], s[2], s[3]);
    } else { // |sum (q-8)*y| over a quad <= 4*8*127: the low/high partials travel as an int16 pair
        int s[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int blk = unit * 4 + b;
            const int yl = a.q32[blk * 8 + u], yh = a.q32[blk * 8 + 4 + u];
            const int sl = dot4((int)(wq[b] & M), yl, 0) - 8 * dot4(0x01010101, yl, 0);
            const int sh = dot4((int)((wq[b] >> 4) & M), yh, 0) - 8 * dot4(0x01010101, yh, 0);
            s[b] = (sl & 0xffff) | (sh << 16);
        }
        return make_int4(s[0], s[1], s[2], s[3]);
    }
}

// hd: Q4_K d|dmin, Q8_0 / Q4_0 the four fp16 block scales of the unit
template <int WT>
__device__ __forceinline__ void rec_chain(const typename RecOf<WT>::T rc, const uint2 hd, const int unit, const LAct a,
                                          float &acc0, float &acc1, float &accm) {
    if constexpr (WT == PS_Q4_K) {
        const float yd   = __uint_as_float(hd.y); // == a.d[unit], read by the caller together with the records
        const float d    = __fmul_rn(yd, ps_h2f((uint16_t)(hd.x & 0xffff)));
        const float dmin = __fmul_rn(-yd, ps_h2f((uint16_t)(hd.x >> 16)));
        acc0 = __fmaf_rn(d, (float)rc.x, acc0);
        accm = __fmaf_rn(dmin, (float)rc.y, accm); // lanes u >= 4: not an acc_m lane, never read
    } else {
        const int sv[4] = {rc.x, rc.y, rc.z, rc.w};
        const uint16_t dh[4] = {(uint16_t)(hd.x & 0xffff), (uint16_t)(hd.x >> 16), (uint16_t)(hd.y & 0xffff), (uint16_t)(hd.y >> 16)};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const float d = __fmul_rn(ps_h2f(dh[b]), a.d[unit * 4 + b]);
            if constexpr (WT == PS_Q8_0) {
                acc0 = __fmaf_rn(d, (float)sv[b], acc0);
            } else {
                acc0 = __fmaf_rn(d, (float)(int)(int16_t)(sv[b] & 0xffff), acc0);
                acc1 = __fmaf_rn(d, (float)(sv[b] >> 16), acc1);
            }
        }
    }
}

// ---- shared by the producer / chain-wave kernels with a Q8_K activation (k_gemv4.hip, k_gemvk.hip)
// silu_hadamard (src/backend/ggml/ggml.cpp:115-129) with glibc's expf table read from LDS: a table lookup in global /
// constant memory is a vector-memory round trip (> 1 us behind the weight stream) on the chain wave's critical path
__device__ __forceinline__ float g4_silu_mul(float g, float u, const uint64_t *tab) {
    float val = g;
    val       = __fmul_rn(val, __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-val, tab))));
    return __fmul_rn(val, u);
}

// Q8_K quantization of one 256-element tile held 4 values per lane (quantize_row_q8_K_ref, ggml-quants.c:3799-3835), the
// prologue's version of ps_quantize_tile: every tile is full (K % 256 == 0), no 16-sums, and the scale / 32-sums are
// written by every lane of their group (same value, same address) instead of behind exec-mask branches -- the prologue is
// issue-bound on a single wave per tile, so instructions are what it costs.
__device__ __forceinline__ void g4_quantize_tile(const float v[4], const int e, const int t, int8_t *qs, float *d, int *bs32, const bool live = true) {
    // straight-line on purpose (no branch on the all-zero tile, `live` guards only the stores): a wave quantizes several
    // independent tiles back to back and the scheduler can only interleave their dependent chains inside one basic block
    int q[4];
    const float am   = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    const float amax = wave_max_dpp(am); // max is order-independent: exact
    // the first element (index order) with the largest |x| decides the sign of iscale
    const unsigned long long hits = __ballot(am == amax);
    const float mine = fabsf(v[0]) == amax ? v[0] : fabsf(v[1]) == amax ? v[1] : fabsf(v[2]) == amax ? v[2] : v[3];
    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), __ffsll((long long)hits) - 1));
    const bool zero = amax == 0.f; // (an all-zero tile: quants 0, d 0 — quantize_row_q8_K's early-out)
    const float iscale = zero ? 0.f : __fdiv_rn(-127.f, mx);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = min(127, __float2int_rn(__fmul_rn(iscale, v[i])));
    const float dd = zero ? 0.f : __fdiv_rn(1.0f, iscale);
    int s = q[0] + q[1] + q[2] + q[3];
    s += dpp_i<0xB1>(s); s += dpp_i<0x4E>(s); s += dpp_i<0x141>(s); // all 8 lanes of a 32-element group hold its sum
    if (live) { // quad-major inside the tile (unit_rec<.., QT>): dword (e / 4) % 64 = g * 8 + u goes to u * 8 + g
        const int dw = (e >> 2) & 63, eq = (e & ~255) + (((dw & 7) << 3) | (dw >> 3)) * 4;
        *(uint32_t *)(qs + eq) = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
        d[t] = dd;
        bs32[e >> 5] = s;
    }
}

} // namespace
