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
#include "ps_quant_dev.h"

namespace {

struct QuantArgs {
    const float *x, *x2, *w;
    float eps;
    int64_t K;
    int8_t *qs;
    float *d;
    int16_t *bs16;
    _Float16 *qf; // Q8_K batches: fragment-major fp16 copy for k_gemm4k.hip, or null
    uint8_t *mf; // ... and the tile-major copy of the column metadata (ps_act::mf), with qf
};

// MODE 1 (RMSNorm needs the whole row): one workgroup per row.
template <int VDT, int TPW>
__global__ __launch_bounds__(256) void quantize_norm_kernel(QuantArgs a) {
    __shared__ double red[8];
    const int64_t row = blockIdx.x, K = a.K;
    const int64_t nblk = K / (VDT == PS_Q8_0 ? 32 : 256);
    ps_quantize_row_wg<VDT, 1, TPW>(a.x + row * K, a.w, a.eps, K, a.qs + row * K, a.d + row * nblk, a.bs16 + row * (K / 16), red,
                                    VDT == PS_Q8_K ? a.qf : nullptr, row, VDT == PS_Q8_K ? a.mf : nullptr);
}
// the same with one wave per tile (K <= 4096: up to 16 waves): the row's tiles are quantized side by side instead of four per wave
// in turn -- a batch row is a latency chain (load, sum of squares, barrier, quantize), not a bandwidth problem
template <int VDT>
__global__ __launch_bounds__(1024) void quantize_norm_wide_kernel(QuantArgs a) {
    __shared__ double red[16];
    const int64_t row = blockIdx.x, K = a.K;
    const int64_t nblk = K / (VDT == PS_Q8_0 ? 32 : 256);
    ps_quantize_row_wg<VDT, 1, 1>(a.x + row * K, a.w, a.eps, K, a.qs + row * K, a.d + row * nblk, a.bs16 + row * (K / 16), red,
                                  VDT == PS_Q8_K ? a.qf : nullptr, row, VDT == PS_Q8_K ? a.mf : nullptr);
}
// MODE 0 / 2: blocks are independent -> one wave per 256-element tile, grid (tiles/4, rows)
template <int VDT, int MODE>
__global__ __launch_bounds__(256) void quantize_tiles_kernel(QuantArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.y, K = a.K, t = (int64_t)blockIdx.x * 4 + wave, e = t * 256 + lane * 4;
    if (t * 256 >= K) return;
    const int64_t nblk = K / (VDT == PS_Q8_0 ? 32 : 256);
    const bool live = e < K;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float4 xv = *(const float4 *)(a.x + row * K + e);
        v[0] = xv.x; v[1] = xv.y; v[2] = xv.z; v[3] = xv.w;
        if (MODE == 2) {
            const float4 uv = *(const float4 *)(a.x2 + row * K + e);
            v[0] = ps_silu_mul(xv.x, uv.x); v[1] = ps_silu_mul(xv.y, uv.y);
            v[2] = ps_silu_mul(xv.z, uv.z); v[3] = ps_silu_mul(xv.w, uv.w);
        }
    }
    ps_quantize_tile<VDT>(v, live, e, t, a.qs + row * K, a.d + row * nblk, a.bs16 + row * (K / 16), nullptr,
                          VDT == PS_Q8_K ? a.qf : nullptr, row, K / 256, VDT == PS_Q8_K ? a.mf : nullptr);
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

// GGUF blocks -> lane-major repack (ps_internal.h).  One thread per output dword.
__global__ void repack_q4_K_kernel(const uint8_t *raw, int64_t N, int64_t nsb, uint32_t *qs, uint4 *hdr) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // dword index in qs
    const int64_t ng = (N + 7) / 8;
    if (o >= ng * nsb * 256) return;
    const int j = (int)(o & 3), u = (int)((o >> 2) & 7), r = (int)((o >> 5) & 7);
    const int64_t gs = o >> 8, sb = gs % nsb, g = gs / nsb, row = g * 8 + r;
    uint32_t v = 0;
    if (row < N) {
        const uint8_t *blk = raw + (row * nsb + sb) * 144;
        v = *(const uint32_t *)(blk + 16 + j * 32 + u * 4);
        if (j == 0 && u == 0) hdr[gs * 8 + r] = *(const uint4 *)blk;
    } else if (j == 0 && u == 0) {
        hdr[gs * 8 + r] = make_uint4(0, 0, 0, 0);
    }
    qs[o] = v;
}
__global__ void repack_q8_0_kernel(const uint8_t *raw, int64_t N, int64_t nb, int64_t nu, uint32_t *qs, uint16_t *d) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ng = (N + 7) / 8;
    if (o >= ng * nu * 256) return;
    const int bl = (int)(o & 3), u = (int)((o >> 2) & 7), r = (int)((o >> 5) & 7);
    const int64_t gs = o >> 8, b4 = gs % nu, g = gs / nu, row = g * 8 + r, blk = b4 * 4 + bl;
    uint32_t v = 0;
    uint16_t dv = 0;
    if (row < N && blk < nb) {
        const uint8_t *b = raw + (row * nb + blk) * 34;
        const uint16_t *s = (const uint16_t *)(b + 2 + u * 4);
        v  = (uint32_t)s[0] | ((uint32_t)s[1] << 16);
        dv = *(const uint16_t *)b;
    }
    qs[o] = v;
    if (u == 0) d[(gs * 8 + r) * 4 + bl] = dv;
}
__global__ void repack_q4_0_kernel(const uint8_t *raw, int64_t N, int64_t nb, int64_t nu, uint32_t *qs, uint16_t *d) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ng = (N + 15) / 16;
    if (o >= ng * nu * 256) return;
    const int bl = (int)(o & 3), u = (int)((o >> 2) & 3), r = (int)((o >> 4) & 15);
    const int64_t gs = o >> 8, b4 = gs % nu, g = gs / nu, row = g * 16 + r, blk = b4 * 4 + bl;
    uint32_t v = 0;
    uint16_t dv = 0;
    if (row < N && blk < nb) {
        const uint8_t *b = raw + (row * nb + blk) * 18;
        const uint16_t *s = (const uint16_t *)(b + 2 + u * 4);
        v  = (uint32_t)s[0] | ((uint32_t)s[1] << 16);
        dv = *(const uint16_t *)b;
    }
    qs[o] = v;
    if (u == 0) d[(gs * 16 + r) * 4 + bl] = dv;
}
__global__ void repack_q6_K_kernel(const uint8_t *raw, int64_t nblk, uint8_t *ql, uint8_t *qh, uint8_t *sc,
                                   uint8_t *d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // 2-byte piece index, 105 per block
    if (i >= nblk * 105) return;
    const int64_t b  = i / 105;
    const int p      = (int)(i % 105);
    const uint16_t v = ((const uint16_t *)raw)[i];
    if (p < 64) { // ql dword (k*8 + u) = bytes k*32 + 4u.. of the block -> lane-major dword u*4 + k
        const int sd = p >> 1, k = sd >> 3, u = sd & 7;
        ((uint16_t *)ql)[b * 64 + (u * 4 + k) * 2 + (p & 1)] = v;
    } else if (p < 96) { // qh dword (j*8 + u) -> u*2 + j
        const int q = p - 64, sd = q >> 1, j = sd >> 3, u = sd & 7;
        ((uint16_t *)qh)[b * 32 + (u * 2 + j) * 2 + (q & 1)] = v;
    }
    else if (p < 104) ((uint16_t *)sc)[b * 8 + (p - 96)] = v;
    else ((uint16_t *)d)[b] = v;
}

__global__ void repack_q5_K_kernel(const uint8_t *raw, int64_t nblk, uint8_t *qs, uint8_t *qh, uint8_t *sc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // dword index, 44 per 176-byte block
    if (i >= nblk * 44) return;
    const int64_t b  = i / 44;
    const int p      = (int)(i % 44);
    const uint32_t v = ((const uint32_t *)raw)[i];
    if (p < 4) ((uint32_t *)sc)[b * 4 + p] = v;        // {d, dmin}, scales[12]
    else if (p < 12) ((uint32_t *)qh)[b * 8 + (p - 4)] = v; // qh dword u = bytes 4u..
    else { // qs dword (jj * 8 + u) = bytes 32 jj + 4u.. -> lane-major dword u * 4 + jj
        const int q = p - 12, jj = q >> 3, u = q & 7;
        ((uint32_t *)qs)[b * 32 + u * 4 + jj] = v;
    }
}

} // namespace

void psk_quantize_act(hipStream_t st, int vdt, int mode, const float *x, const float *x2, const float *w, float eps,
                      int64_t K, int64_t rows, ps_act out) {
    const bool frag = vdt == PS_Q8_K && rows >= ps_gemm4k_min_cols() && K % 1024 == 0; // exactly when psk_gemm4k takes the batch
    QuantArgs a{x, x2, w, eps, K, out.qs, out.d, out.bs16, frag ? out.qf : nullptr, frag ? out.mf : nullptr};
    if (mode == 1) {
        dim3 g((unsigned)rows), b(256);
        const int64_t tpw = ((K + 255) / 256 + 3) / 4;
        static const bool wide_off = getenv("PS_NO_QNORM_WIDE") != nullptr;
        if (!wide_off && rows >= 2 && K % 256 == 0 && K / 256 >= 8 && K / 256 <= 16) { // batches: one wave per tile
            dim3 bw((unsigned)(K / 256 * 64));
            if (vdt == PS_Q8_0) hipLaunchKernelGGL((quantize_norm_wide_kernel<PS_Q8_0>), g, bw, 0, st, a);
            else hipLaunchKernelGGL((quantize_norm_wide_kernel<PS_Q8_K>), g, bw, 0, st, a);
            return;
        }
#define LN(V) do { if (tpw <= 4) hipLaunchKernelGGL((quantize_norm_kernel<V, 4>), g, b, 0, st, a); \
                   else if (tpw <= 8) hipLaunchKernelGGL((quantize_norm_kernel<V, 8>), g, b, 0, st, a); \
                   else hipLaunchKernelGGL((quantize_norm_kernel<V, 16>), g, b, 0, st, a); } while (0)
        if (vdt == PS_Q8_0) LN(PS_Q8_0); else LN(PS_Q8_K);
#undef LN
        return;
    }
    dim3 g((unsigned)(((K + 255) / 256 + 3) / 4), (unsigned)rows), b(256);
    if (vdt == PS_Q8_0) {
        if (mode == 0) hipLaunchKernelGGL((quantize_tiles_kernel<PS_Q8_0, 0>), g, b, 0, st, a);
        else hipLaunchKernelGGL((quantize_tiles_kernel<PS_Q8_0, 2>), g, b, 0, st, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((quantize_tiles_kernel<PS_Q8_K, 0>), g, b, 0, st, a);
        else hipLaunchKernelGGL((quantize_tiles_kernel<PS_Q8_K, 2>), g, b, 0, st, a);
    }
}

void psk_pack_act_blocks(hipStream_t st, int vdt, ps_act in, int64_t K, int64_t rows, void *out_blocks) {
    const int64_t nb = ((vdt == PS_Q8_0) ? K / 32 : K / 256) * rows;
    hipLaunchKernelGGL(pack_act_blocks_kernel, dim3((unsigned)((nb + 127) / 128)), dim3(128), 0, st, vdt, in.qs, in.d,
                       in.bs16, K, rows, (uint8_t *)out_blocks);
}

void psk_repack_weight(hipStream_t st, int dtype, const uint8_t *raw, int64_t K, int64_t N, ps_weight *w) {
    const int T = 256;
    if (dtype == PS_Q4_0 || dtype == PS_Q8_0) {
        const int64_t nb = K / 32, nu = (K + 127) / 128, rg = dtype == PS_Q4_0 ? 16 : 8, ng = (N + rg - 1) / rg;
        const int64_t n = ng * nu * 256;
        if (dtype == PS_Q4_0)
            hipLaunchKernelGGL(repack_q4_0_kernel, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, st, raw, N, nb, nu, (uint32_t *)w->qs, (uint16_t *)w->aux);
        else
            hipLaunchKernelGGL(repack_q8_0_kernel, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, st, raw, N, nb, nu, (uint32_t *)w->qs, (uint16_t *)w->aux);
    } else if (dtype == PS_Q4_K) {
        const int64_t nsb = K / 256, ng = (N + 7) / 8, n = ng * nsb * 256;
        hipLaunchKernelGGL(repack_q4_K_kernel, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, st, raw, N, nsb, (uint32_t *)w->qs, (uint4 *)w->aux);
    } else if (dtype == PS_Q5_K) {
        const int64_t nb = N * (K / 256);
        hipLaunchKernelGGL(repack_q5_K_kernel, dim3((unsigned)((nb * 44 + T - 1) / T)), dim3(T), 0, st, raw, nb, w->qs, w->qh, w->sc);
    } else if (dtype == PS_Q6_K) {
        const int64_t nb = N * (K / 256);
        hipLaunchKernelGGL(repack_q6_K_kernel, dim3((unsigned)((nb * 105 + T - 1) / T)), dim3(T), 0, st, raw, nb, w->qs,
                           w->qh, w->sc, w->aux);
    }
}
