// Activation-row quantizer shared by the stand-alone kernels (k_quant.hip) and the fused GEMV prologue
// (k_gemv.hip).  Bit-exact restatement of quantize_row_q8_0 (AVX2 branch, libs/ggml/src/ggml-quants.c:957-1039)
// and quantize_row_q8_K (:3799-3835), with optional producers:
//   MODE 1  RMSNorm  y = x * (w * scale)                  (ggml.c:12667-12720, :2442-2470)
//   MODE 2  SiLU*up  val = g * (1/(1+expf(-g))) * u        (src/backend/ggml/ggml.cpp:115-129)
// A wave owns a tile of 256 consecutive elements, lane l elements 4l..4l+3 (one coalesced float4 per lane).
// A Q8_0 block (32 elements) is 8 consecutive lanes, a bsums group (16) is 4 lanes, a Q8_K block is the whole
// wave: every reduction is a wave shuffle.  Output pointers may be global or LDS.
#pragma once
#include "ps_dev.h"
#include "ps_expf.h"

__device__ __forceinline__ float ps_silu_mul(float g, float u) {
    float val = g;
    val       = __fmul_rn(val, __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-val))));
    return __fmul_rn(val, u);
}

// quantize the 4 values of this lane (elements e..e+3 of tile t of the row); wave-collective
template <int VDT>
__device__ __forceinline__ void ps_quantize_tile(const float v[4], bool live, int64_t e, int64_t t, int8_t *qs, float *d,
                                                 int16_t *bs16, int *bs32 = nullptr, // bs32: optional int sums of 32 (two bs16)
                                                 _Float16 *qf = nullptr, int64_t col = 0, int64_t nsb = 0, // qf: fragment-major fp16 copy (ps_act::qf)
                                                 uint8_t *mf = nullptr) { // mf: tile-major copy of the column metadata (ps_act::mf)
    const int lane = threadIdx.x & 63;
    int q[4];
    if (VDT == PS_Q8_0) {
        float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        amax       = group8_max_dpp(amax);
        const float dd = __fdiv_rn(amax, 127.f);
        const float id = (amax != 0.0f) ? __fdiv_rn(127.f, amax) : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = __float2int_rn(__fmul_rn(v[i], id)); // round-half-even
        if (live && (lane & 7) == 0) d[e / 32] = ps_h2f(ps_f2h(dd));
    } else { // Q8_K: the first element (index order) with the strictly largest |x| decides the sign of iscale
        float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        const float amax = wave_max_dpp(am);                          // max is order-independent: exact
        // smallest index attaining amax = lowest lane holding such an element (ballot + find-first-set, scalar) and, in
        // that lane, the first of its four elements
        const unsigned long long hits = __ballot(am == amax);
        if (amax == 0.f) {
            q[0] = q[1] = q[2] = q[3] = 0;
            if (live && lane == 0) { d[t] = 0.f; if (mf) *(float *)(mf + ((col >> 4) * nsb + t) * 576 + (col & 15) * 4) = 0.f; }
        } else {
            const float mine = fabsf(v[0]) == amax ? v[0] : fabsf(v[1]) == amax ? v[1] : fabsf(v[2]) == amax ? v[2] : v[3];
            const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), __ffsll((long long)hits) - 1));
            const float iscale = __fdiv_rn(-127.f, mx);
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = min(127, __float2int_rn(__fmul_rn(iscale, v[i])));
            if (live && lane == 0) {
                const float dd = __fdiv_rn(1.0f, iscale);
                d[t] = dd;
                if (mf) *(float *)(mf + ((col >> 4) * nsb + t) * 576 + (col & 15) * 4) = dd;
            }
        }
    }
    const int s16 = group4_sum_i_dpp(q[0] + q[1] + q[2] + q[3]);
    if (live) {
        const uint32_t packed = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) |
                                ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
        *(uint32_t *)(qs + e) = packed;
        if (qf) { // lane = (sub-block g, quad u) of super-block t: [16-column tile][t][u][kb * 16 + col % 16][half][e0, e2, e1, e3]
            const int g = lane >> 3, u = lane & 7;
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 hv = {(_Float16)(float)q[0], (_Float16)(float)q[2], (_Float16)(float)q[1], (_Float16)(float)q[3]};
            *(h4 *)((char *)qf + ((((col >> 4) * nsb + t) << 13) + (u << 10) + ((((g >> 1) << 4) + (col & 15)) << 4) + ((g & 1) << 3))) = hv;
        }
        if ((lane & 3) == 0) {
            bs16[e / 16] = (int16_t)s16;
            if (VDT == PS_Q8_K && mf) *(_Float16 *)(mf + ((col >> 4) * nsb + t) * 576 + 64 + (col & 15) * 32 + (lane >> 2) * 2) = (_Float16)(float)s16; // |s16| <= 16 * 127: exact in fp16
        }
    }
    if (bs32) {
        const int s32 = s16 + dpp_i<0x104>(s16); // lane & 7 == 0: + the next four lanes' 16-sum
        if (live && (lane & 7) == 0) bs32[e / 32] = s32;
    }
}

// Whole row by one workgroup of NW waves, split in two phases so that a caller can put OTHER loads in flight
// between them (vmcnt retires in order: loads issued after these do not delay the wait for these):
//   ps_qrow_load     issue every global load of the row (and of the norm weights) into registers
//   ps_qrow_compute  RMSNorm (MODE 1) + quantization from those registers; ends with __syncthreads()
// Requires K <= nw*TPW*256 (one tile slot per (wave, i)).  `red`: __shared__ double[16].
template <int MODE, int TPW>
__device__ __forceinline__ void ps_qrow_load(const float *x, const float *w, int64_t K, float4 (&xv)[TPW], float4 (&wv)[TPW], int nwl = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nwl ? nwl : (int)(blockDim.x >> 6); // nwl: only the first nwl waves take tiles
    const int64_t n_tiles = (K + 255) / 256;
#pragma unroll
    for (int i = 0; i < TPW; i++) {
        const int64_t t = wave < nw ? wave + (int64_t)i * nw : n_tiles, e = t * 256 + lane * 4;
        const bool in = t < n_tiles && e < K;
        // never a branch around a load (hipcc waits vmcnt(0) behind one): an out-of-range slot re-reads the row's first 16
        // bytes and is zeroed by a select
        const int64_t ec = in ? e : 0;
        const float4 xl = *(const float4 *)(x + ec);
        xv[i] = in ? xl : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1) {
            const float4 wl = *(const float4 *)(w + ec);
            wv[i] = in ? wl : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
// the same tile map, activation only, fetched with cache-bypassing (agent-scope relaxed atomic) 64-bit loads: the row was
// written by other workgroups of the SAME kernel (chained mat-vec phases) and must not be served from a stale L2 line
template <int TPW>
__device__ __forceinline__ void ps_qrow_load_coh(const float *x, int64_t K, float4 (&xv)[TPW], int nwl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_tiles = (K + 255) / 256;
#pragma unroll
    for (int i = 0; i < TPW; i++) {
        const int64_t t = wave < nwl ? wave + (int64_t)i * nwl : n_tiles, e = t * 256 + lane * 4;
        unsigned long long lo = 0, hi = 0;
        if (t < n_tiles && e < K) { // K % 4 == 0 on this path
            lo = __hip_atomic_load((const unsigned long long *)(x + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hi = __hip_atomic_load((const unsigned long long *)(x + e + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        xv[i] = make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
    }
}
struct PsNoMark { __device__ __forceinline__ void operator()(int, float) const {} }; // timeline hook: (event, value the event depends on)
template <int VDT, int MODE, int TPW, class Mark = PsNoMark>
__device__ __forceinline__ void ps_qrow_compute(const float4 (&xv)[TPW], const float4 (&wv)[TPW], float eps, int64_t K, int8_t *qs,
                                                float *d, int16_t *bs16, double *red, int nwl = 0, Mark mk = Mark(), int *bs32 = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, nwt = nwl ? nwl : nw;
    const int64_t n_tiles = (K + 255) / 256;
    float scale = 1.0f;
    if (MODE == 1) {
        // sum += (double)(x*x); mean = sum/ne00; scale = 1/sqrtf(mean + eps)
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            s += (double)__fmul_rn(xv[i].x, xv[i].x);
            s += (double)__fmul_rn(xv[i].y, xv[i].y);
            s += (double)__fmul_rn(xv[i].z, xv[i].z);
            s += (double)__fmul_rn(xv[i].w, xv[i].w);
        }
        mk(24, (float)s); // activation row arrived, squares summed in the lane
        s = wave_sum_d_dpp(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        mk(25, 0.f); // partial sums exchanged
        double tot = 0.0;
        for (int i = 0; i < nw; i++) tot += red[i];
        const float mean = (float)(tot / (double)K);
        scale            = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, eps)));
        mk(26, scale); // scale known
    }
#pragma unroll
    for (int i = 0; i < TPW; i++) {
        const int64_t t = wave < nwt ? wave + (int64_t)i * nwt : n_tiles, e = t * 256 + lane * 4;
        if (t >= n_tiles) continue; // wave-uniform
        const bool live = e < K;
        float v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
        if (MODE == 1 && live) {
            v[0] = __fmul_rn(v[0], __fmul_rn(wv[i].x, scale));
            v[1] = __fmul_rn(v[1], __fmul_rn(wv[i].y, scale));
            v[2] = __fmul_rn(v[2], __fmul_rn(wv[i].z, scale));
            v[3] = __fmul_rn(v[3], __fmul_rn(wv[i].w, scale));
        }
        ps_quantize_tile<VDT>(v, live, e, t, qs, d, bs16, bs32);
    }
    mk(27, 0.f); // tiles quantized
    __syncthreads();
}

// Whole row by one workgroup of NW waves; up to TPW tiles per wave are held in registers so that all global
// loads are in flight together (K <= NW*TPW*256).  MODE 0 plain, 1 RMSNorm.  `red`: __shared__ double[16].
// Ends with __syncthreads().
template <int VDT, int MODE, int TPW>
__device__ __forceinline__ void ps_quantize_row_wg(const float *x, const float *w, float eps, int64_t K, int8_t *qs, float *d,
                                                   int16_t *bs16, double *red, _Float16 *qf = nullptr, int64_t col = 0, uint8_t *mf = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t n_tiles = (K + 255) / 256;
    for (int64_t t0 = 0; t0 < n_tiles; t0 += (int64_t)nw * TPW) { // one trip when K <= nw*TPW*256 (always, for MODE 1)
        float4 xv[TPW], wv[MODE == 1 ? TPW : 1];
#pragma unroll
        for (int i = 0; i < TPW; i++) { // every global load of the row (and of the norm weights) in flight at once
            const int64_t t = t0 + wave + (int64_t)i * nw, e = t * 256 + lane * 4;
            const bool in = t < n_tiles && e < K;
            xv[i] = in ? *(const float4 *)(x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 1) wv[i] = in ? *(const float4 *)(w + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float scale = 1.0f;
        if (MODE == 1) {
            // sum += (double)(x*x); mean = sum/ne00; scale = 1/sqrtf(mean + eps)
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < TPW; i++) {
                s += (double)__fmul_rn(xv[i].x, xv[i].x);
                s += (double)__fmul_rn(xv[i].y, xv[i].y);
                s += (double)__fmul_rn(xv[i].z, xv[i].z);
                s += (double)__fmul_rn(xv[i].w, xv[i].w);
            }
            s = wave_sum_d_dpp(s);
            __syncthreads();
            if (lane == 0) red[wave] = s;
            __syncthreads();
            double tot = 0.0;
            for (int i = 0; i < nw; i++) tot += red[i];
            const float mean = (float)(tot / (double)K);
            scale            = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, eps)));
        }
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            const int64_t t = t0 + wave + (int64_t)i * nw, e = t * 256 + lane * 4;
            if (t >= n_tiles) continue; // wave-uniform
            const bool live = e < K;
            float v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
            if (MODE == 1 && live) {
                v[0] = __fmul_rn(v[0], __fmul_rn(wv[i].x, scale));
                v[1] = __fmul_rn(v[1], __fmul_rn(wv[i].y, scale));
                v[2] = __fmul_rn(v[2], __fmul_rn(wv[i].z, scale));
                v[3] = __fmul_rn(v[3], __fmul_rn(wv[i].w, scale));
            }
            ps_quantize_tile<VDT>(v, live, e, t, qs, d, bs16, nullptr, qf, col, K / 256, mf);
        }
    }
    __syncthreads();
}
