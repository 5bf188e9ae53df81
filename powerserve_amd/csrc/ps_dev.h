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
// ---- the reference's second build (-DPS_CONTRACT; powerserve_amd/build.py build(contract=True) -> lib/libps_hip_contract.so).
// The reference's own CMake sets no fp-contraction flag, so a stock build is GCC's default -ffp-contract=fast: of everything on the hot path
// three scalar places come out different (tests/test_ref_fast.py): the RoPE rotation, the n % 32 leftovers of ggml_vec_dot_f32,
// and Q5_K's summs (not implemented under PS_CONTRACT: ps_hip_weight_upload refuses Q5_K there).  Default: every operation rounds where the C
// source rounds (the -ffp-contract=off build the oracle, the golden vectors and "bit-exact" refer to).
// ---- in-kernel timelines (s_memtime marks read back by ps_hip_debug_timeline, tools/gpu_*timeline*.py) are compiled in only with -DPS_TIMELINE=1
// (powerserve_amd/build.py build(timeline=True) -> lib/libps_hip_timeline.so).  Round 5 measured what the dormant marks cost the library that ships: a
// branch per mark, the timeline pointer's scalar load and wait at the head of every launch, registers -- all mat-vecs of a token 1.343 -> 1.293 ms, decode
// 548 -> 563 tok/s without them (profiles/r05_decode_experiments.txt).
#ifndef PS_TIMELINE
#define PS_TIMELINE 0
#endif
#define PS_TL(ptr) (PS_TIMELINE ? (ptr) : nullptr)
#ifdef PS_CONTRACT
constexpr bool PS_CONTRACT_ON = true;
#else
constexpr bool PS_CONTRACT_ON = false;
#endif
// x0 * c - x1 * s, x0 * s + x1 * c (ggml.c:15455-15456, :15474-15475); contracted: GCC fuses the first product of each with the add
__device__ __forceinline__ void ps_rope_pair(const float x0, const float x1, const float c, const float s, float &ra, float &rb) {
    if (PS_CONTRACT_ON) { ra = __fmaf_rn(x0, c, -__fmul_rn(x1, s)); rb = __fmaf_rn(x0, s, __fmul_rn(x1, c)); }
    else { ra = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s)); rb = __fadd_rn(__fmul_rn(x0, s), __fmul_rn(x1, c)); }
}
// one output of the pair (odd: the second), for epilogues whose lane finishes a single element
__device__ __forceinline__ float ps_rope_one(const float x0, const float x1, const float c, const float s, const bool odd) {
    if (PS_CONTRACT_ON) return odd ? __fmaf_rn(x0, s, __fmul_rn(x1, c)) : __fmaf_rn(x0, c, -__fmul_rn(x1, s));
    return odd ? __fadd_rn(__fmul_rn(x0, s), __fmul_rn(x1, c)) : __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s));
}
// leftover j (0-based) of the nleft = n % 32 scalar steps `sumf += x[i] * y[i]` (ggml.c:2123-2125); contracted: GCC vectorises the loop by 8
// and by 4 without fusing and fuses only the last nleft % 4 steps
__device__ __forceinline__ float ps_dot_left(const float sum, const float x, const float y, const int j, const int nleft) {
    if (PS_CONTRACT_ON && j >= (nleft & ~3)) return __fmaf_rn(x, y, sum);
    return __fadd_rn(sum, __fmul_rn(x, y));
}
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

// ---- DPP (data-parallel primitive) lane exchange: VALU-speed cross-lane moves inside a row of 16 lanes,
//      no LDS-pipe round trip (ds_bpermute ~100 cycles each).  ctrl: 0xB1 quad_perm[1,0,3,2] (xor 1),
//      0x4E quad_perm[2,3,0,1] (xor 2), 0x141 row_half_mirror (lane 7-i), 0x140 row_mirror (lane 15-i),
//      0x101..0x10F row_shl:n (lane i reads lane i+n, 0 beyond the row).
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int)(b & 0xffffffffll)), hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// order-insensitive reductions (max / min / integer sums / double sums that are not order-critical)
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = row16_max(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ int wave_min_i_dpp(int v) {
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}
__device__ __forceinline__ double wave_sum_d_dpp(double v) {
    v += dpp_d<0xB1>(v); v += dpp_d<0x4E>(v); v += dpp_d<0x141>(v); v += dpp_d<0x140>(v);
    const long long b = __double_as_longlong(v);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int l = __builtin_amdgcn_readlane(lo, 16 * k), h = __builtin_amdgcn_readlane(hi, 16 * k);
        r[k] = __longlong_as_double(((long long)h << 32) | (unsigned int)l);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}
__device__ __forceinline__ float group8_max_dpp(float v) { // aligned groups of 8 lanes
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v));
    return v;
}
__device__ __forceinline__ int group4_sum_i_dpp(int v) { // aligned groups of 4 lanes
    v += dpp_i<0xB1>(v); v += dpp_i<0x4E>(v);
    return v;
}

// ---- GGML_F32x8_REDUCE (libs/ggml/src/ggml.c:1354-1371) over the 32 fp32 chains of ggml_vec_dot_f32's AVX build,
//      held by 32 consecutive lanes (c = lane & 31 = accumulator*8 + simd lane): acc0+=acc2, acc1+=acc3 (xor 16),
//      acc0+=acc1 (xor 8), low+high 128 bits (xor 4), two hadds (xor 1, xor 2).  Valid in the lane with c == 0.
//      Only the lane with c == 0 needs the result, so every step is a one-directional fetch (lane c reads lane c + n):
//      c + 16 through v_permlane16_swap (gfx950: swaps the odd 16-lane rows of one operand with the even rows of the other;
//      with both operands = v the second result holds row 1 in row 0 and row 3 in row 2), the rest as DPP row shifts —
//      VALU-speed, no ds_bpermute round trips.  Same partners and the same association as the xor butterfly.
__device__ __forceinline__ float reduce_f32x8x4(float v) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __fadd_rn(v, __uint_as_float(sw[1])); // c < 16: + lane c + 16   (acc0 += acc2, acc1 += acc3)
    v = __fadd_rn(v, dpp_f<0x108>(v));        // c < 8:  + lane c + 8    (acc0 += acc1)
    v = __fadd_rn(v, dpp_f<0x104>(v));        // c < 4:  + lane c + 4    (low + high 128 bits)
    v = __fadd_rn(v, dpp_f<0x101>(v));        // c = 0, 2: + lane c + 1  (hadd)
    v = __fadd_rn(v, dpp_f<0x102>(v));        // c = 0:  + lane 2        (hadd)
    return v;
}

// ---- int8 dot: 4 signed bytes x 4 signed bytes + acc  (v_dot4_i32_i8)
__device__ __forceinline__ int dot4(int a, int b, int acc) {
    return __builtin_amdgcn_sdot4(a, b, acc, false);
}

// 4 independent v_dot4_i32_i8 in the VOP3P form with src2 = 0.  (The builtin only ever selects the accumulating VOP2
// form, v_dot4c, which costs a v_mov 0 per product when nothing is accumulated.)  A DOT result must not be read or
// overwritten by a different VALU instruction within 3 wait states and the hazard recognizer cannot see into an asm
// statement: the trailing s_nop 2 covers the last product, the earlier ones are covered by the products after them.
__device__ __forceinline__ void dot4x4(int (&d)[4], const int a0, const int a1, const int a2, const int a3, const int b0, const int b1,
                                       const int b2, const int b3) {
    asm("v_dot4_i32_i8 %0, %4, %8, 0\n\t"
        "v_dot4_i32_i8 %1, %5, %9, 0\n\t"
        "v_dot4_i32_i8 %2, %6, %10, 0\n\t"
        "v_dot4_i32_i8 %3, %7, %11, 0\n\t"
        "s_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
// (single inline-asm dot4s are not used: DOT results need wait states before their first use on gfx950 and the hazard recognizer
//  cannot see inside an asm statement -- measured: wrong sums)
// two signed 16-bit products + 32-bit accumulator (v_dot2_i32_i16)
typedef short ps_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2_i16(uint32_t a, uint32_t b, int acc) {
    ps_i16x2 va, vb;
    __builtin_memcpy(&va, &a, 4);
    __builtin_memcpy(&vb, &b, 4);
    return __builtin_amdgcn_sdot2(va, vb, acc, false);
}

// ---- streaming (read-once) 16-byte load: non-temporal so weight bytes do not displace L2-resident
//      activations (MI355X guide: nt-weights, -18 % issue->landed latency on decode weight streams)
typedef uint32_t ps_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream16(const void *p) {
    const ps_u32x4 v = __builtin_nontemporal_load((const ps_u32x4 *)p); // global_load_dwordx4 ... nt
    return make_uint4(v.x, v.y, v.z, v.w);
}

// (No inline-asm global loads: the compiler cannot see that an asm load's destination is still in flight, copies such
//  registers freely and reads garbage; the mat-vec instead keeps every load unconditional so that the compiler's own
//  s_waitcnt counts stay exact.)
typedef uint32_t ps_u32x2 __attribute__((ext_vector_type(2)));

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

// N values at once, in place: x[i] = ggml_v_expf(x[i] - mx).  The same operation sequence per value as ps_v_expf; what changes is the control
// flow.  ps_v_expf's rare |n| > 126 path (overflow / underflow scaling; x - max <= 0 gets there only below -87, i.e. masked -inf logits) is a
// per-lane branch, and called value after value it fences each value's dependent chain of ~12 fp operations off from the next: eight exponentials
// ran as eight latency chains in a row (the soft-max phases are bound by exactly that: profiles/r03_attention_timeline.txt "exp" 0.98 us).  Here
// the common path of all N values is straight-line code (the chains interleave) and the fix-up sits behind ONE wave-uniform test.
template <int N>
__device__ __forceinline__ void ps_v_expf_n(float (&x)[N], const float mx) {
    const float r = 0x1.8p23f;
    float nn[N], jj[N];
    uint32_t ee[N];
    bool slow = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float xi = __fsub_rn(x[i], mx);
        const float z = __fmaf_rn(xi, 0x1.715476p+0f, r);
        const float n = __fsub_rn(z, r);
        const float b = __fmaf_rn(-n, 0x1.7f7d1cp-20f, __fmaf_rn(-n, 0x1.62e4p-1f, xi));
        const uint32_t e = __float_as_uint(z) << 23;
        const float k = __uint_as_float(e + 0x3f800000u);
        const float u = __fmul_rn(b, b);
        const float j = __fmaf_rn(__fmaf_rn(__fmaf_rn(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, __fmaf_rn(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, __fmul_rn(0x1.ffffecp-1f, b));
        x[i] = __fmaf_rn(j, k, k);
        nn[i] = n; jj[i] = j; ee[i] = e;
        slow = slow || fabsf(n) > 126.0f;
    }
    if (__builtin_amdgcn_ballot_w64(slow) != 0) { // (wave-uniform: rarely taken)
#pragma unroll
        for (int i = 0; i < N; i++) {
            const float n = nn[i];
            const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
            const float s1 = __uint_as_float(g + 0x7f000000u);
            const float s2 = __uint_as_float(ee[i] - g);
            const float big = __fmul_rn(s1, s1), mid = __fmul_rn(__fmaf_rn(s2, jj[i], s2), s1);
            x[i] = fabsf(n) > 126.0f ? (fabsf(n) > 192.0f ? big : mid) : x[i];
        }
    }
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
