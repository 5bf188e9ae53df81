// Device code shared by the single-column Q4_K kernels (k_gemv4.hip: gemv4_kernel; k_qkvattn.hip: qkv_attn_kernel): the per-unit integer work of a
// producer lane with the row headers expanded once per chunk, and the 1 KiB LDS-DMA pull.
#pragma once
#include "ps_gemv_dev.h"

namespace {
// A row's header of one super-block, expanded ONCE per chunk (by the lane that loaded it) into what the eight lanes of the
// row need for each unit, so that they do not all unpack the same 6-bit fields again (the producers are VALU-bound:
// ~62 instructions per unit and lane, a third of them header work).  64 bytes in LDS:
//   [0..15]  sc16[j] = {scale[2j], scale[2j+1]} as int16 pairs (the v_dot2_i32_i16 operand)
//   [16..47] mins as int32 pairs {min[2v], min[2v+1]}, v = 0..3 (lane u takes pair u & 3)
//   [48..55] d, dmin as fp32
constexpr int G4_HX = 64;
__device__ __forceinline__ void g4_expand_header(const ps_u32x4 hc, char *dst) {
    const uint32_t sc03 = hc.y & 0x3f3f3f3fu, sc47 = (hc.w & 0x0f0f0f0fu) | (((hc.y >> 6) & 0x03030303u) << 4);
    const uint32_t mn03 = hc.z & 0x3f3f3f3fu, mn47 = ((hc.w >> 4) & 0x0f0f0f0fu) | (((hc.z >> 6) & 0x03030303u) << 4);
    *(uint4 *)dst = make_uint4(__builtin_amdgcn_perm(0u, sc03, 0x0c010c00u), __builtin_amdgcn_perm(0u, sc03, 0x0c030c02u),
                               __builtin_amdgcn_perm(0u, sc47, 0x0c010c00u), __builtin_amdgcn_perm(0u, sc47, 0x0c030c02u));
    *(uint4 *)(dst + 16) = make_uint4(mn03 & 0xff, (mn03 >> 8) & 0xff, (mn03 >> 16) & 0xff, mn03 >> 24);
    *(uint4 *)(dst + 32) = make_uint4(mn47 & 0xff, (mn47 >> 8) & 0xff, (mn47 >> 16) & 0xff, mn47 >> 24);
    *(float2 *)(dst + 48) = make_float2(ps_h2f((uint16_t)(hc.x & 0xffff)), ps_h2f((uint16_t)(hc.x >> 16)));
}
// one unit against the activation column (unit_rec<PS_Q4_K, QT> of ps_gemv_dev.h with the header work taken out):
// returns the record {d * yd, (float)sumi, -dmin * yd, (float)(mins . bsums)}
__device__ __forceinline__ float4 g4_unit(const ps_u32x4 q, const char *hx, const int unit, const int u, const LAct a) {
    constexpr uint32_t M = 0x0F0F0F0Fu;
    const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
    const int v = u & 3;
    const int4 y0 = *(const int4 *)(a.q32 + unit * 64 + u * 8), y1 = *(const int4 *)(a.q32 + unit * 64 + u * 8 + 4);
    const int2 bs = *(const int2 *)(a.bs32 + unit * 8 + 2 * v);
    const uint4 sc16 = *(const uint4 *)hx;
    const int2 mp = *(const int2 *)(hx + 16 + v * 8);
    const float2 dd = *(const float2 *)(hx + 48);
    const float yd = a.d[unit];
    int dlo[4], dhi[4]; // the eight quad dots as plain v_dot4 (no zeroed accumulators)
    dot4x4(dlo, (int)(wq[0] & M), (int)(wq[1] & M), (int)(wq[2] & M), (int)(wq[3] & M), y0.x, y0.z, y1.x, y1.z);
    dot4x4(dhi, (int)((wq[0] >> 4) & M), (int)((wq[1] >> 4) & M), (int)((wq[2] >> 4) & M), (int)((wq[3] >> 4) & M), y0.y, y0.w, y1.y, y1.w);
    const uint32_t scv[4] = {sc16.x, sc16.y, sc16.z, sc16.w};
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) // |dot4| <= 4*15*127 fits int16: {dl, dh} meet their scale pair in one v_dot2_i32_i16 (exact)
        s = dot2_i16(__builtin_amdgcn_perm((uint32_t)dhi[j], (uint32_t)dlo[j], 0x05040100u), scv[j], s);
    const int pr = __mul24(mp.x, bs.x) + __mul24(mp.y, bs.y);
    return make_float4(__fmul_rn(yd, dd.x), (float)s, __fmul_rn(-yd, dd.y), (float)pr);
}

// the same with the activation operands already in registers (gemv4_kernel's YS): y0 / y1 the lane's 32 quants, bs its pair of 32-sums
__device__ __forceinline__ float4 g4_unit_y(const ps_u32x4 q, const char *hx, const int u, const int4 y0, const int4 y1, const int2 bs, const float yd) {
    constexpr uint32_t M = 0x0F0F0F0Fu;
    const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
    const int v = u & 3;
    const uint4 sc16 = *(const uint4 *)hx;
    const int2 mp = *(const int2 *)(hx + 16 + v * 8);
    const float2 dd = *(const float2 *)(hx + 48);
    int dlo[4], dhi[4];
    dot4x4(dlo, (int)(wq[0] & M), (int)(wq[1] & M), (int)(wq[2] & M), (int)(wq[3] & M), y0.x, y0.z, y1.x, y1.z);
    dot4x4(dhi, (int)((wq[0] >> 4) & M), (int)((wq[1] >> 4) & M), (int)((wq[2] >> 4) & M), (int)((wq[3] >> 4) & M), y0.y, y0.w, y1.y, y1.w);
    const uint32_t scv[4] = {sc16.x, sc16.y, sc16.z, sc16.w};
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) s = dot2_i16(__builtin_amdgcn_perm((uint32_t)dhi[j], (uint32_t)dlo[j], 0x05040100u), scv[j], s);
    const int pr = __mul24(mp.x, bs.x) + __mul24(mp.y, bs.y);
    return make_float4(__fmul_rn(yd, dd.x), (float)s, __fmul_rn(-yd, dd.y), (float)pr);
}

// one wave-instruction pulls 1 KiB through the L2 into an LDS dump (global_load_lds_dwordx4: lane l lands at M0 + 16 l; no destination register, so nothing
// to wait for and nothing the compiler could reuse too early): the chain wave's L2 prefetch below
__device__ __forceinline__ unsigned g4_lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p; }
__device__ __forceinline__ void g4_pull_nt(const uint8_t *q, const unsigned lds_dst) { // (the same with the non-temporal hint)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(q), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void g4_pull(const uint8_t *q, const unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(q), "s"(lds_dst) : "memory");
}

} // namespace
