// Device code shared by the quantized mat-vec kernels (k_gemv.hip: gemv / gemv1 / gemm8*, k_gemv4.hip, k_gemvb.hip, k_gemvk.hip):
// weight-type traits, the LDS image of an activation column, the per-unit integer work (unit_dot) and hsum_float_8 (row_reduce).  See k_gemv.hip's header for the numerics contract.
#pragma once
// (round 6) what a decode step STREAMS and reads once per token should not cycle through the L2s and the memory-side cache in front of what is reused (activations,
// exchanges): the quants carry the non-temporal hint since round 2; the cached K rows / V channels since round 6 (k_qkvattn.hip KVS, k_attn.hip: +2.1 % decode on the 8B shape).
// G4_HDR_NT: the Q4_K row headers of the FUSED Q / K / V + attention launch as well (same box, rocprofv3 per launch, profiles/r06_hdr_nt_kernel_times.txt: 20.62 -> 20.33 us;
// the mat-vec launches of their own do not move -- 14.49 -> 14.54 us for gate / up -- and keep plain header loads).  NOT the Q6_K / Q5_K scales (gemvk: the eight lanes of a row
// share one 16-byte piece; 8B Q4_K_M 530 -> 519 tok/s with the hint) nor the Q4_0 / Q8_0 block scales (gemvb: no difference): profiles/r06_nt_other_ab.txt
#ifndef G4_HDR_NT
#define G4_HDR_NT 1
#endif
#include <type_traits>
#include "ps_dev.h"
#include "ps_internal.h"
#include "ps_ops.h"
#include "ps_quant_dev.h"

// the result rows of the Q4_0 / Q8_0 / Q6_K / Q5_K mat-vecs stored write-through like gemv4's (k_gemv4.hip G4_OUT_WT); A/B: -DPS_OUT_WT=0
#ifndef PS_OUT_WT
#define PS_OUT_WT 1
#endif
__device__ __forceinline__ void ps_out_wt(float *p, const float v) { if (PS_OUT_WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; }

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

// TWO tiles per wave-instruction (round 4): lanes 0..31 hold tile 2 tp (eight consecutive elements each), lanes 32..63 tile 2 tp + 1.  The
// prologue of the mat-vec kernels is bound by vector-instruction issue (K = 14336: 56 tiles x ~150 instructions on one CU = 3.8 us), and most of a
// tile's instructions are the same for every lane -- the two correctly rounded divisions (-127 / max, 1 / iscale), the reduction steps, the
// first-maximum search: with a tile per half wave each of them serves two tiles.  Same arithmetic per element as g4_quantize_tile, same LDS image.
// (A four-tile form -- a tile per row of 16 lanes, ~50 instructions per tile -- measured slower: 524 vs 535 tok/s, profiles/r04_gemv_variants.txt.)
// v[8]: elements e0 .. e0 + 7 of the row, e0 = tp * 512 + 8 * lane.
__device__ __forceinline__ void g4_quantize_pair(const float v[8], const int e0, const int tp, int8_t *qs, float *d, int *bs32, const bool live, const int n_units = 1 << 30) {
    const int lane = threadIdx.x & 63;
    const bool hi = lane >= 32;
    float am = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
    // maximum of each half wave (max is order-independent: exact): rows of 16 lanes by DPP, the two rows of a half through scalar registers
    const float rm = row16_max(am);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rm), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rm), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rm), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rm), 48));
    const float amax = hi ? fmaxf(r2, r3) : fmaxf(r0, r1);
    // the first element (index order) with the largest |x| decides the sign of iscale: the lowest lane of the half with a hit, its lowest element
    const unsigned long long hits = __ballot(am == amax);
    float mine = v[7];
#pragma unroll
    for (int i = 6; i >= 0; i--) mine = fabsf(v[i]) == amax ? v[i] : mine;
    const unsigned lo32 = (unsigned)hits, hi32 = (unsigned)(hits >> 32);
    const int la = lo32 ? __ffs((int)lo32) - 1 : 0, lb = hi32 ? 32 + __ffs((int)hi32) - 1 : 32; // (an all-NaN half has no hit: any lane, the result is NaN either way)
    const float ma = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), la)), mb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), lb));
    const float mx = hi ? mb : ma;
    const bool zero = amax == 0.f; // (an all-zero tile: quants 0, d 0 -- quantize_row_q8_K's early-out)
    const float iscale = zero ? 0.f : __fdiv_rn(-127.f, mx);
    int q[8];
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = min(127, __float2int_rn(__fmul_rn(iscale, v[i])));
    const float dd = zero ? 0.f : __fdiv_rn(1.0f, iscale);
    int s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    s += dpp_i<0xB1>(s); s += dpp_i<0x4E>(s); // the four lanes of a 32-element group hold its sum
    const int t = 2 * tp + (hi ? 1 : 0);
    if (live && t < n_units) { // quad-major inside the tile: dword (e / 4) % 64 = g * 8 + u goes to u * 8 + g  (n_units: an odd row's last pair has one tile)
        const int dw = (e0 >> 2) & 63, base = e0 & ~255;
        const uint32_t p0 = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
        const uint32_t p1 = (uint32_t)(q[4] & 0xff) | ((uint32_t)(q[5] & 0xff) << 8) | ((uint32_t)(q[6] & 0xff) << 16) | ((uint32_t)(q[7] & 0xff) << 24);
        *(uint32_t *)(qs + base + (((dw & 7) << 3) | (dw >> 3)) * 4) = p0;             // quad dw (even: dw & 7 in {0, 2, 4, 6})
        *(uint32_t *)(qs + base + ((((dw + 1) & 7) << 3) | (dw >> 3)) * 4) = p1;       // quad dw + 1: the same group of eight, the next u
        d[t] = dd;
        bs32[e0 >> 5] = s;
    }
}

} // namespace
