// Activation quantization for the integer dot products (gfx950).
//
// Restates, bit-exactly, what powerserve_compute_forward_mul_mat does to the F32 activation before any
// dot product (libs/ggml/src/ggml.c:13502-13530): quantize_row_q8_0 for Q4_0/Q8_0 weights (AVX2 branch,
// ggml-quants.c:957-1039) and quantize_row_q8_K for Q4_K/Q6_K weights (ggml-quants.c:3799-3835).
// Optional fused producers: RMSNorm (ggml.c:12667-12720) and SiLU*up (backend/ggml/ggml.cpp:115-129).
//
// Mapping: one workgroup (4 waves) per activation row; a wave owns tiles of 256 consecutive elements,
// lane l owns elements 4l..4l+3 of the tile (one coalesced float4 per lane, 1 KiB per wave).  A Q8_0
// block (32 elements) is therefore 8 consecutive lanes, a bsums group (16) is 4 lanes, a Q8_K block is
// the whole wave: all reductions are wave shuffles, no LDS except the RMSNorm row sum.
//
// Built with -ffp-contract=off: every multiply/add below rounds exactly where the C source of the
// reference rounds.
#include "ps_dev.h"
#include "ps_internal.h"

namespace {

struct QuantArgs {
    const float *x, *x2, *w;
    float eps;
    int64_t K;
    int8_t *qs;
    float *d;
    int16_t *bs16;
};

__device__ __forceinline__ float produce(int mode, float xv, float x2v, float wv, float scale) {
    if (mode == 1) return __fmul_rn(xv, __fmul_rn(wv, scale)); // y = x * (w * scale)   (ggml.c:2466)
    if (mode == 2) {                                            // silu_hadamard        (ggml.cpp:122-127)
        float val = xv;
        val       = __fmul_rn(val, __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-val))));
        return __fmul_rn(val, x2v);
    }
    return xv;
}

template <int VDT, int MODE>
__global__ __launch_bounds__(256) void quantize_act_kernel(QuantArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.x, K = a.K;
    const float *x  = a.x + row * K;
    const float *x2 = a.x2 ? a.x2 + row * K : nullptr;
    const int64_t n_tiles = (K + 255) / 256;

    float scale = 1.0f;
    if (MODE == 1) {
        // sum += (double)(x*x); mean = sum/ne00; scale = 1/sqrtf(mean + eps)   (ggml.c:12698-12707)
        __shared__ double red[4];
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
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const double tot = (red[0] + red[1]) + (red[2] + red[3]);
        const float mean = (float)(tot / (double)K);
        scale            = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, a.eps)));
    }

    for (int64_t t = wave; t < n_tiles; t += 4) {
        const int64_t e  = t * 256 + lane * 4;
        const bool live  = e < K;
        float v[4]       = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const float4 xv = *(const float4 *)(x + e);
            float4 x2v      = make_float4(0, 0, 0, 0), wv = make_float4(0, 0, 0, 0);
            if (MODE == 2) x2v = *(const float4 *)(x2 + e);
            if (MODE == 1) wv = *(const float4 *)(a.w + e);
            v[0] = produce(MODE, xv.x, x2v.x, wv.x, scale);
            v[1] = produce(MODE, xv.y, x2v.y, wv.y, scale);
            v[2] = produce(MODE, xv.z, x2v.z, wv.z, scale);
            v[3] = produce(MODE, xv.w, x2v.w, wv.w, scale);
        }
        int q[4];
        if (VDT == PS_Q8_0) {
            float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            amax       = group_max<8>(amax);
            const float d  = __fdiv_rn(amax, 127.f);
            const float id = (amax != 0.0f) ? __fdiv_rn(127.f, amax) : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = __float2int_rn(__fmul_rn(v[i], id)); // round-half-even
            if (live && (lane & 7) == 0) a.d[row * (K / 32) + e / 32] = ps_h2f(ps_f2h(d));
        } else { // Q8_K
            // first element (in index order) with the strictly largest |x| decides the sign of iscale
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
                if (live && lane == 0) a.d[row * (K / 256) + t] = 0.f;
            } else {
                const float iscale = __fdiv_rn(-127.f, mx);
#pragma unroll
                for (int i = 0; i < 4; i++) q[i] = min(127, __float2int_rn(__fmul_rn(iscale, v[i])));
                if (live && lane == 0) a.d[row * (K / 256) + t] = __fdiv_rn(1.0f, iscale);
            }
        }
        const int s16 = group_sum_i<4>(q[0] + q[1] + q[2] + q[3]);
        if (live) {
            const uint32_t packed = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) |
                                    ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
            *(uint32_t *)(a.qs + row * K + e) = packed;
            if ((lane & 3) == 0) a.bs16[row * (K / 16) + e / 16] = (int16_t)s16;
        }
    }
}

// SoA activation -> GGUF block layout (block_q8_0 34 B / block_q8_K 292 B), for parity tests of the
// quantizer through the C-ABI.
__global__ void pack_act_blocks_kernel(int vdt, const int8_t *qs, const float *d, const int16_t *bs16, int64_t K,
                                       int64_t rows, uint8_t *out) {
    const int64_t nb_row = (vdt == PS_Q8_0) ? K / 32 : K / 256;
    const int64_t b      = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb_row * rows) return;
    const int64_t row = b / nb_row, ib = b % nb_row;
    if (vdt == PS_Q8_0) {
        uint8_t *o       = out + b * 34;
        const uint16_t h = ps_f2h(d[row * nb_row + ib]);
        o[0]             = (uint8_t)(h & 0xff);
        o[1]             = (uint8_t)(h >> 8);
        for (int i = 0; i < 32; i++) o[2 + i] = (uint8_t)qs[row * K + ib * 32 + i];
    } else {
        uint8_t *o     = out + b * 292;
        const float dv = d[row * nb_row + ib];
        memcpy(o, &dv, 4);
        for (int i = 0; i < 256; i++) o[4 + i] = (uint8_t)qs[row * K + ib * 256 + i];
        for (int i = 0; i < 16; i++) {
            const int16_t s = bs16[row * (K / 16) + ib * 16 + i];
            memcpy(o + 260 + 2 * i, &s, 2);
        }
    }
}

// GGUF blocks -> backend SoA (see ps_internal.h).  One thread per 16-byte piece / per block.
__global__ void repack_q4_0_kernel(const uint8_t *raw, int64_t nblk, uint8_t *qs, uint8_t *d) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const uint16_t *s = (const uint16_t *)(raw + b * 18);
    ((uint16_t *)d)[b] = s[0];
    uint16_t *o        = (uint16_t *)(qs + b * 16);
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = s[1 + i];
}
__global__ void repack_q8_0_kernel(const uint8_t *raw, int64_t nblk, uint8_t *qs, uint8_t *d) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const uint16_t *s = (const uint16_t *)(raw + b * 34);
    ((uint16_t *)d)[b] = s[0];
    uint16_t *o        = (uint16_t *)(qs + b * 32);
#pragma unroll
    for (int i = 0; i < 16; i++) o[i] = s[1 + i];
}
__global__ void repack_q4_K_kernel(const uint4 *raw, int64_t nblk, uint4 *qs, uint4 *hdr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // 16-byte piece index
    if (i >= nblk * 9) return;
    const int64_t b = i / 9;
    const int c     = (int)(i % 9);
    const uint4 v   = raw[i];
    if (c == 0) hdr[b] = v; else qs[b * 8 + (c - 1)] = v;
}
__global__ void repack_q6_K_kernel(const uint8_t *raw, int64_t nblk, uint8_t *ql, uint8_t *qh, uint8_t *sc,
                                   uint8_t *d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // 2-byte piece index, 105 per block
    if (i >= nblk * 105) return;
    const int64_t b  = i / 105;
    const int p      = (int)(i % 105);
    const uint16_t v = ((const uint16_t *)raw)[i];
    if (p < 64) ((uint16_t *)ql)[b * 64 + p] = v;
    else if (p < 96) ((uint16_t *)qh)[b * 32 + (p - 64)] = v;
    else if (p < 104) ((uint16_t *)sc)[b * 8 + (p - 96)] = v;
    else ((uint16_t *)d)[b] = v;
}

} // namespace

void psk_quantize_act(hipStream_t st, int vdt, int mode, const float *x, const float *x2, const float *w, float eps,
                      int64_t K, int64_t rows, ps_act out) {
    QuantArgs a{x, x2, w, eps, K, out.qs, out.d, out.bs16};
    dim3 g((unsigned)rows), b(256);
#define LAUNCH(V, M) hipLaunchKernelGGL((quantize_act_kernel<V, M>), g, b, 0, st, a)
    if (vdt == PS_Q8_0) {
        if (mode == 0) LAUNCH(PS_Q8_0, 0); else if (mode == 1) LAUNCH(PS_Q8_0, 1); else LAUNCH(PS_Q8_0, 2);
    } else {
        if (mode == 0) LAUNCH(PS_Q8_K, 0); else if (mode == 1) LAUNCH(PS_Q8_K, 1); else LAUNCH(PS_Q8_K, 2);
    }
#undef LAUNCH
}

void psk_pack_act_blocks(hipStream_t st, int vdt, ps_act in, int64_t K, int64_t rows, void *out_blocks) {
    const int64_t nb = ((vdt == PS_Q8_0) ? K / 32 : K / 256) * rows;
    hipLaunchKernelGGL(pack_act_blocks_kernel, dim3((unsigned)((nb + 127) / 128)), dim3(128), 0, st, vdt, in.qs, in.d,
                       in.bs16, K, rows, (uint8_t *)out_blocks);
}

void psk_repack_weight(hipStream_t st, int dtype, const uint8_t *raw, int64_t K, int64_t N, ps_weight *w) {
    const int T = 256;
    if (dtype == PS_Q4_0) {
        const int64_t nb = N * (K / 32);
        hipLaunchKernelGGL(repack_q4_0_kernel, dim3((unsigned)((nb + T - 1) / T)), dim3(T), 0, st, raw, nb, w->qs, w->aux);
    } else if (dtype == PS_Q8_0) {
        const int64_t nb = N * (K / 32);
        hipLaunchKernelGGL(repack_q8_0_kernel, dim3((unsigned)((nb + T - 1) / T)), dim3(T), 0, st, raw, nb, w->qs, w->aux);
    } else if (dtype == PS_Q4_K) {
        const int64_t nb = N * (K / 256);
        hipLaunchKernelGGL(repack_q4_K_kernel, dim3((unsigned)((nb * 9 + T - 1) / T)), dim3(T), 0, st, (const uint4 *)raw,
                           nb, (uint4 *)w->qs, (uint4 *)w->aux);
    } else if (dtype == PS_Q6_K) {
        const int64_t nb = N * (K / 256);
        hipLaunchKernelGGL(repack_q6_K_kernel, dim3((unsigned)((nb * 105 + T - 1) / T)), dim3(T), 0, st, raw, nb, w->qs,
                           w->qh, w->sc, w->aux);
    }
}
