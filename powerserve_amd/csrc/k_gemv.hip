// Quantized weight x quantized activation mat-vec for decode (gfx950, HBM-bound).
//
// Replaces the per-(row, col) vec_dot loop of powerserve_compute_forward_mul_mat
// (libs/ggml/src/ggml.c:13578-13647, one_chunk :13344-13432) with the integer block dots of
//   ggml_vec_dot_q4_0_q8_0 (ggml-quants.c:3935), ggml_vec_dot_q8_0_q8_0 (:5532), ggml_vec_dot_q4_K_q8_K (:7727).
// Integer parts are exact; the fp32 scale-accumulate uses the same products (dx*dy per block,
// d*y.d / dmin*y.d per super-block) with a different summation order (<= ~1e-6 relative).
//
// Design (memory-bound: every weight byte is read exactly once, nothing else touches HBM):
//   * weights live in a structure-of-arrays repack (ps_internal.h): a wave reads a row's quant plane as
//     fully coalesced 16 B/lane = 1 KiB "chunks" with non-temporal loads; TB chunks per row and R rows are
//     put in flight before any arithmetic (no LDS round trip for weights — GEMV operands are used once).
//   * the quantized activation (a few KB) is staged once per workgroup into LDS and re-read per chunk with
//     ds_read_b128; it is shared by every row the workgroup streams.
//   * one wave = one output row group; lanes hold 4x v_dot4_i32_i8 partial sums, the row result is a
//     64-lane shuffle reduction.  Up to three matrices share one launch (QKV, gate/up) and the epilogue
//     applies bias / residual add / SiLU*up so those never become separate kernels.
#include "ps_dev.h"
#include "ps_internal.h"

namespace {

struct GemvW {
    const uint8_t *qs;
    const uint8_t *aux;
    float *out;
    const float *bias;
    int64_t N, ldo;
};

struct GemvParams {
    GemvW w[3];
    int n_w;
    int64_t rows_total, K, bs;
    const float *residual;
    int64_t col_bytes; // LDS bytes per activation column (16-B multiple)
    const int8_t *aq;
    const float *ad;
    const int16_t *abs16;
};

template <int WT> struct WTraits;
template <> struct WTraits<PS_Q4_0> { static constexpr int EPC = 2048, BLK = 32;  };  // elements per 1 KiB chunk
template <> struct WTraits<PS_Q4_K> { static constexpr int EPC = 2048, BLK = 256; };
template <> struct WTraits<PS_Q8_0> { static constexpr int EPC = 1024, BLK = 32;  };

// LDS image of one activation column
struct LAct {
    const int8_t *qs;
    const float *d;
    const int *bs32;
};

__device__ __forceinline__ float silu_mul(float g, float u) { // backend/ggml/ggml.cpp:122-127
    float val = g;
    val       = __fmul_rn(val, __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-val))));
    return __fmul_rn(val, u);
}

// one 16-byte piece of a row's quant plane against one activation column
template <int WT>
__device__ __forceinline__ void chunk_dot(const uint4 q, const uint4 h, const float dw, const int lane, const int t,
                                          const LAct a, float &facc, float &macc) {
    constexpr uint32_t M = 0x0F0F0F0Fu;
    if (WT == PS_Q4_K) {
        const int c = lane & 7, sb = t * 8 + (lane >> 3);
        const int e0 = sb * 256 + (c >> 1) * 64 + (c & 1) * 16;
        const int4 yl = *(const int4 *)(a.qs + e0);
        const int4 yh = *(const int4 *)(a.qs + e0 + 32);
        int dl = dot4((int)(q.x & M), yl.x, 0);
        dl     = dot4((int)(q.y & M), yl.y, dl);
        dl     = dot4((int)(q.z & M), yl.z, dl);
        dl     = dot4((int)(q.w & M), yl.w, dl);
        int dh = dot4((int)((q.x >> 4) & M), yh.x, 0);
        dh     = dot4((int)((q.y >> 4) & M), yh.y, dh);
        dh     = dot4((int)((q.z >> 4) & M), yh.z, dh);
        dh     = dot4((int)((q.w >> 4) & M), yh.w, dh);
        int sc0, sc1, mc, tmp;
        ps_scale_min_k4(2 * (c >> 1), h.y, h.z, h.w, sc0, tmp);
        ps_scale_min_k4(2 * (c >> 1) + 1, h.y, h.z, h.w, sc1, tmp);
        ps_scale_min_k4(c, h.y, h.z, h.w, tmp, mc);
        const float yd   = a.d[sb];
        const float d    = __fmul_rn(yd, ps_h2f((uint16_t)(h.x & 0xffff)));
        const float dmin = __fmul_rn(-yd, ps_h2f((uint16_t)(h.x >> 16)));
        facc = __fmaf_rn(d, (float)(sc0 * dl + sc1 * dh), facc);
        macc = __fmaf_rn(dmin, (float)(mc * a.bs32[sb * 8 + c]), macc);
    } else if (WT == PS_Q4_0) {
        const int bi = t * 64 + lane;
        const int4 yl = *(const int4 *)(a.qs + bi * 32);
        const int4 yh = *(const int4 *)(a.qs + bi * 32 + 16);
        int s = dot4((int)(q.x & M), yl.x, 0);
        s     = dot4((int)(q.y & M), yl.y, s);
        s     = dot4((int)(q.z & M), yl.z, s);
        s     = dot4((int)(q.w & M), yl.w, s);
        s     = dot4((int)((q.x >> 4) & M), yh.x, s);
        s     = dot4((int)((q.y >> 4) & M), yh.y, s);
        s     = dot4((int)((q.z >> 4) & M), yh.z, s);
        s     = dot4((int)((q.w >> 4) & M), yh.w, s);
        s -= 8 * a.bs32[bi]; // sum (q-8)*y = sum q*y - 8*sum y
        facc = __fmaf_rn(__fmul_rn(dw, a.d[bi]), (float)s, facc);
    } else { // Q8_0
        const int eo = t * 1024 + lane * 16;
        const int4 y = *(const int4 *)(a.qs + eo);
        int s = dot4((int)q.x, y.x, 0);
        s     = dot4((int)q.y, y.y, s);
        s     = dot4((int)q.z, y.z, s);
        s     = dot4((int)q.w, y.w, s);
        facc = __fmaf_rn(__fmul_rn(dw, a.d[eo >> 5]), (float)s, facc);
    }
}

// MODE 0: out = y (+bias) (+residual);  MODE 1: out[0] = silu(y_w0) * y_w1 (R == 2, rows paired)
template <int WT, int TB, int R, int BS, int MODE>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvParams p) {
    using TR = WTraits<WT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t K = p.K;
    const int nblk  = (int)(K / TR::BLK);            // scale blocks per row
    const int nb32  = (int)(K / 32);
    const size_t col_bytes = (size_t)p.col_bytes;

    // ---- stage the quantized activation column(s) into LDS
    for (int col = 0; col < BS; col++) {
        char *base = smem + col * col_bytes;
        int8_t *lq = (int8_t *)base;
        float *ld  = (float *)(base + (K + 15) / 16 * 16);
        int *lb    = (int *)(ld + nblk);
        const bool on = col < p.bs;
        for (int64_t i = threadIdx.x * 16; i < K; i += 256 * 16)
            *(int4 *)(lq + i) = on ? *(const int4 *)(p.aq + col * K + i) : make_int4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < nblk; i += 256) ld[i] = on ? p.ad[col * nblk + i] : 0.f;
        for (int i = threadIdx.x; i < nb32; i += 256) {
            const int16_t *b = p.abs16 + col * (K / 16) + 2 * i;
            lb[i]            = on ? (int)b[0] + (int)b[1] : 0;
        }
    }
    __syncthreads();

    const int Tn          = (int)((K + TR::EPC - 1) / TR::EPC); // 1 KiB chunks per row
    const int64_t row_qb  = (WT == PS_Q8_0) ? K : K / 2;         // quant-plane bytes per row
    const int64_t n_groups = (MODE == 1) ? p.w[0].N : (p.rows_total + R - 1) / R;

    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < n_groups; g += (int64_t)gridDim.x * 4) {
        const uint8_t *qrow[R];
        const uint8_t *arow[R];
        int wi[R];
        int64_t lr[R];
        bool rv[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            int64_t row = (MODE == 1) ? g : g * R + r;
            int i       = (MODE == 1) ? r : 0;
            rv[r]       = (MODE == 1) ? true : row < p.rows_total;
            if (MODE == 0) {
                if (!rv[r]) row = 0;
                if (p.n_w > 1 && row >= p.w[0].N) { row -= p.w[0].N; i = 1; }
                if (p.n_w > 2 && i == 1 && row >= p.w[1].N) { row -= p.w[1].N; i = 2; }
            }
            wi[r]   = i;
            lr[r]   = row;
            qrow[r] = p.w[i].qs + row * row_qb;
            arow[r] = p.w[i].aux + row * (int64_t)nblk * ((WT == PS_Q4_K) ? 16 : 2);
        }
        float facc[R][BS], macc[R][BS];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < BS; c++) { facc[r][c] = 0.f; macc[r][c] = 0.f; }

        for (int t0 = 0; t0 < Tn; t0 += TB) {
            uint4 q[R][TB], h[R][TB];
            float dw[R][TB];
            bool ok[TB];
            // ---- put every load of this batch in flight first
#pragma unroll
            for (int tt = 0; tt < TB; tt++) {
                const int t        = t0 + tt;
                const int64_t boff = (int64_t)t * 1024 + lane * 16;
                ok[tt]             = (t < Tn) && (boff < row_qb);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    q[r][tt] = make_uint4(0, 0, 0, 0);
                    h[r][tt] = make_uint4(0, 0, 0, 0);
                    dw[r][tt] = 0.f;
                    if (ok[tt]) {
                        q[r][tt] = ld_stream16(qrow[r] + boff);
                        if (WT == PS_Q4_K) h[r][tt] = *(const uint4 *)(arow[r] + (size_t)(t * 8 + (lane >> 3)) * 16);
                        else if (WT == PS_Q4_0) dw[r][tt] = ps_h2f(*(const uint16_t *)(arow[r] + (size_t)(t * 64 + lane) * 2));
                        else dw[r][tt] = ps_h2f(*(const uint16_t *)(arow[r] + (size_t)(t * 32 + (lane >> 1)) * 2));
                    }
                }
            }
            // ---- integer dots against the LDS-resident activation
#pragma unroll
            for (int tt = 0; tt < TB; tt++) {
                if (!ok[tt]) continue;
#pragma unroll
                for (int c = 0; c < BS; c++) {
                    const char *base = smem + c * col_bytes;
                    LAct a;
                    a.qs   = (const int8_t *)base;
                    a.d    = (const float *)(base + (K + 15) / 16 * 16);
                    a.bs32 = (const int *)(a.d + nblk);
#pragma unroll
                    for (int r = 0; r < R; r++) chunk_dot<WT>(q[r][tt], h[r][tt], dw[r][tt], lane, t0 + tt, a, facc[r][c], macc[r][c]);
                }
            }
        }
        // ---- row results + epilogue
#pragma unroll
        for (int c = 0; c < BS; c++) {
            float y[R];
#pragma unroll
            for (int r = 0; r < R; r++) y[r] = wave_sum(facc[r][c]) + wave_sum(macc[r][c]);
            if (lane == 0 && c < p.bs) {
                if (MODE == 1) {
                    p.w[0].out[c * p.w[0].ldo + g] = silu_mul(y[0], y[R - 1]);
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!rv[r]) continue;
                        const GemvW &W = p.w[wi[r]];
                        float v        = y[r];
                        if (W.bias) v = __fadd_rn(v, W.bias[lr[r]]);
                        if (p.residual && wi[r] == 0) v = __fadd_rn(p.residual[c * W.ldo + lr[r]], v);
                        W.out[c * W.ldo + lr[r]] = v;
                    }
                }
            }
        }
    }
}

template <int WT, int TB, int R, int BS, int MODE>
void launch_one(hipStream_t st, int n_cu, const GemvParams &p, size_t smem) {
    const int64_t n_groups = (MODE == 1) ? p.w[0].N : (p.rows_total + R - 1) / R;
    int64_t grid           = (n_groups + 3) / 4;
    const int64_t cap      = (int64_t)n_cu * (smem > 36 * 1024 ? 2 : 4);
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    static bool attr_set = false;
    if (!attr_set && smem > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)gemv_kernel<WT, TB, R, BS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemv_kernel<WT, TB, R, BS, MODE>), dim3((unsigned)grid), dim3(256), smem, st, p);
}

template <int WT, int BS, int MODE>
void launch_tb(hipStream_t st, int n_cu, const GemvParams &p, size_t smem, int Tn, bool pair_rows) {
    // R = 2 when rows are short (few chunks) or paired; TB = chunks kept in flight per row
    if (MODE == 1 || pair_rows) {
        if (Tn <= 1) launch_one<WT, 1, 2, BS, MODE>(st, n_cu, p, smem);
        else if (Tn == 2) launch_one<WT, 2, 2, BS, MODE>(st, n_cu, p, smem);
        else launch_one<WT, 4, 2, BS, MODE>(st, n_cu, p, smem);
    } else {
        if (Tn <= 1) launch_one<WT, 1, 1, BS, MODE>(st, n_cu, p, smem);
        else if (Tn == 2) launch_one<WT, 2, 1, BS, MODE>(st, n_cu, p, smem);
        else if (Tn == 7) launch_one<WT, 7, 1, BS, MODE>(st, n_cu, p, smem);
        else launch_one<WT, 4, 1, BS, MODE>(st, n_cu, p, smem);
    }
}

template <int WT>
int launch_wt(hipStream_t st, int n_cu, const GemvParams &p, size_t col_bytes, int mode) {
    const int Tn = (int)((p.K + WTraits<WT>::EPC - 1) / WTraits<WT>::EPC);
    bool even    = true;
    for (int i = 0; i < p.n_w; i++) even = even && (p.w[i].N % 2 == 0);
    const bool pair_rows = even && Tn <= 2;
    if (p.bs == 1) {
        if (mode == 1) launch_tb<WT, 1, 1>(st, n_cu, p, col_bytes, Tn, true);
        else launch_tb<WT, 1, 0>(st, n_cu, p, col_bytes, Tn, pair_rows);
    } else if (p.bs <= 4) {
        if (mode == 1) launch_tb<WT, 4, 1>(st, n_cu, p, col_bytes * 4, Tn, true);
        else launch_tb<WT, 4, 0>(st, n_cu, p, col_bytes * 4, Tn, false);
    } else {
        return 3;
    }
    return 0;
}

} // namespace

// bs <= 4 columns per launch; larger batches go through the MFMA GEMM (k_gemm.hip) or are split by the caller.
int psk_gemv(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int vdt, int64_t K, int64_t bs) {
    GemvParams p{};
    p.n_w        = a.n_w;
    p.rows_total = 0;
    p.K          = K;
    p.bs         = bs;
    p.residual   = a.residual;
    p.aq         = act.qs;
    p.ad         = act.d;
    p.abs16      = act.bs16;
    const int wt = a.w[0]->dtype;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != wt || a.w[i]->K != K) return 4;
        p.w[i] = GemvW{a.w[i]->qs, a.w[i]->aux, a.out[i], a.bias[i], a.w[i]->N, a.ldo[i]};
        p.rows_total += a.w[i]->N;
    }
    (void)vdt;
    const int64_t blk      = (wt == PS_Q4_K) ? 256 : 32;
    const size_t col_bytes = ((size_t)((K + 15) / 16 * 16) + (size_t)(K / blk) * 4 + (size_t)(K / 32) * 4 + 15) / 16 * 16;
    p.col_bytes            = (int64_t)col_bytes;
    const int mode         = a.silu_pair ? 1 : 0;
    if (mode == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N)) return 5;
    switch (wt) {
    case PS_Q4_0: return launch_wt<PS_Q4_0>(st, n_cu, p, col_bytes, mode);
    case PS_Q8_0: return launch_wt<PS_Q8_0>(st, n_cu, p, col_bytes, mode);
    case PS_Q4_K: return launch_wt<PS_Q4_K>(st, n_cu, p, col_bytes, mode);
    }
    return 6;
}
