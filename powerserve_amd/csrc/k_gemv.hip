// Quantized weight x quantized activation mat-vec (gfx950) — HBM-bound, and BIT-EXACT with the reference.
//
// Replaces the per-(row, col) vec_dot loop of powerserve_compute_forward_mul_mat
// (libs/ggml/src/ggml.c:13578-13647, one_chunk :13344-13432).  The reference's x86 build runs the AVX2
// kernels ggml_vec_dot_q4_0_q8_0 (ggml-quants.c:4205-4228), ggml_vec_dot_q8_0_q8_0 (:5761-5782) and
// ggml_vec_dot_q4_K_q8_K (:7809-7873).  Each keeps EIGHT fp32 accumulator lanes: lane u sums, per 32-element
// block, the integer partial over elements 4u..4u+3 ("quad u"), converts it to float and does
//     acc[u] = fma(d_block, (float)partial[u], acc[u])            sequentially over the blocks of the row,
// then reduces the 8 lanes with hsum_float_8 (ggml-quants.c:62-68).  A different fp32 summation order changes
// the last bits of every mat-mul output, and because the next op re-quantizes activations to int8 a 1-ulp
// difference can flip a quant and grow to 1e-2 in the logits.  So this kernel reproduces the order exactly:
//
//   * one GPU lane = one (weight row r, AVX lane u).  The weights are repacked at upload (ps_internal.h)
//     so that the 16 bytes a lane needs for one "unit" (a Q4_K super-block / four Q4_0 or Q8_0 blocks) are
//     contiguous and a wavefront's 64 x 16 B load is one fully coalesced 1 KiB piece of 8 (16) rows.
//     Every weight byte is still read exactly once, with non-temporal loads, no LDS round trip.
//   * the lane computes quad partials with v_dot4_i32_i8 (exact), the per-block scale product exactly as
//     the reference (d = dx*dy), and runs its own fma chain in block order; hsum_float_8's association is a
//     xor-4, xor-2, xor-1 butterfly over the 8 lanes of a row.  Result == reference, bit for bit.
//   * the activation row is quantized ONCE PER WORKGROUP in the prologue (RMSNorm / plain / pre-quantized),
//     bit-exactly (ps_quant_dev.h), into LDS; it is re-read per unit with ds_read_b32 broadcast across rows.
//   * up to three matrices share a launch (QKV, gate+up) and the epilogue applies bias / residual / SiLU*up.
#include "ps_dev.h"
#include "ps_internal.h"
#include "ps_quant_dev.h"

namespace {

struct GemvW {
    const uint8_t *qs;
    const uint8_t *aux;
    float *out;
    const float *bias;
    int64_t N, ldo, n_groups;
};

struct GemvParams {
    GemvW w[3];
    int n_w;
    int64_t groups_total, K, bs;
    const float *residual;
    int64_t col_bytes; // LDS bytes per activation column (16-B multiple)
    // prologue: 0 = activation already quantized (aq/ad/abs16), 1 = rmsnorm(x, nw, eps), 2 = quantize(x)
    const float *x, *nw;
    float eps;
    const int8_t *aq;
    const float *ad;
    const int16_t *abs16;
};

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
    if (WT == PS_Q4_0) {
        float r = __fadd_rn(acc1, acc0); // a[k+4] + a[k]
        r = __fadd_rn(r, __shfl_xor(r, 2, 64));
        r = __fadd_rn(r, __shfl_xor(r, 1, 64));
        return r;
    }
    float r = __fadd_rn(acc0, __shfl_xor(acc0, 4, 64));
    r = __fadd_rn(r, __shfl_xor(r, 2, 64));
    r = __fadd_rn(r, __shfl_xor(r, 1, 64));
    if (WT == PS_Q4_K) {
        float m = __fadd_rn(accm, __shfl_xor(accm, 2, 64)); // (m0+m2), (m1+m3)
        m = __fadd_rn(m, __shfl_xor(m, 1, 64));
        r = __fadd_rn(r, m);
    }
    return r;
}

// EPI 0: out = y (+bias) (+residual).   EPI 1: out[0] = silu(y_w0) * y_w1 (same row of w[0] and w[1]).
template <int WT, int BS, int EPI, int PRO>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvParams p) {
    using TR = WTraits<WT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t K = p.K;
    const int Kp   = (int)((K + TR::UNIT - 1) / TR::UNIT * TR::UNIT); // padded to whole units; pad region is zero
    const int nblk = Kp / TR::BLK, nb32 = Kp / 32, nb16 = Kp / 16;
    const int nblk_k = (int)(K / TR::BLK), nb16_k = (int)(K / 16);

    // ---- activation column(s) -> LDS: [int8 q[Kp]] [float d[nblk]] [int bs32[nb32]] [int16 bs16[nb16] scratch]
    for (int col = 0; col < BS; col++) {
        char *base   = smem + col * p.col_bytes;
        int8_t *lq   = (int8_t *)base;
        float *ld    = (float *)(base + Kp);
        int *lb      = (int *)(ld + nblk);
        int16_t *l16 = (int16_t *)(lb + nb32);
        if (col < p.bs) {
            if (Kp != K) { // zero the padding blocks (their weights are zero too: fma(0, s, acc) leaves acc unchanged)
                for (int i = (int)K + threadIdx.x * 4; i < Kp; i += 256 * 4) *(int *)(lq + i) = 0;
                for (int i = nblk_k + threadIdx.x; i < nblk; i += 256) ld[i] = 0.f;
                for (int i = nb16_k + threadIdx.x; i < nb16; i += 256) l16[i] = 0;
            }
            if (PRO == 0) {
                for (int64_t i = threadIdx.x * 16; i < K; i += 256 * 16) *(int4 *)(lq + i) = *(const int4 *)(p.aq + col * K + i);
                for (int i = threadIdx.x; i < nblk_k; i += 256) ld[i] = p.ad[col * nblk_k + i];
                for (int i = threadIdx.x; i < nb16_k; i += 256) l16[i] = p.abs16[col * nb16_k + i];
                __syncthreads();
            } else {
                ps_quantize_row_wg<TR::VDT, PRO == 1 ? 1 : 0, 16>(p.x + col * K, p.nw, p.eps, K, lq, ld, l16, red);
            }
            for (int i = threadIdx.x; i < nb32; i += 256) lb[i] = (int)l16[2 * i] + (int)l16[2 * i + 1];
        } else {
            for (int i = threadIdx.x * 4; i < Kp; i += 256 * 4) *(int *)(lq + i) = 0;
            for (int i = threadIdx.x; i < nblk; i += 256) ld[i] = 0.f;
            for (int i = threadIdx.x; i < nb32; i += 256) lb[i] = 0;
        }
    }
    __syncthreads();

    const int r = (WT == PS_Q4_0) ? (lane >> 2) : (lane >> 3);
    const int u = (WT == PS_Q4_0) ? (lane & 3) : (lane & 7);
    const int n_units        = (int)((K + TR::UNIT - 1) / TR::UNIT);
    const int64_t unit_bytes = 1024;                                   // quant-plane bytes per unit per row group
    const int64_t aux_unit   = (int64_t)TR::RG * (WT == PS_Q4_K ? 16 : 8);
    const int64_t n_tasks    = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    constexpr int UNR = 4;

    for (int64_t task = (int64_t)blockIdx.x * 4 + wave; task < n_tasks; task += (int64_t)gridDim.x * 4) {
        float yres[BS][EPI == 1 ? 2 : 1];
        int wi = 0;
        int64_t grp = task;
#pragma unroll
        for (int pass = 0; pass < (EPI == 1 ? 2 : 1); pass++) {
            if (EPI == 1) {
                wi = pass;
            } else {
                if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
                if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
            }
            const GemvW &W    = p.w[wi];
            const uint8_t *qg = W.qs + grp * n_units * unit_bytes + (int64_t)lane * 16;
            const uint8_t *ag = W.aux + grp * n_units * aux_unit + (int64_t)r * (WT == PS_Q4_K ? 16 : 8);
            float acc0[BS], acc1[BS], accm[BS];
#pragma unroll
            for (int c = 0; c < BS; c++) { acc0[c] = 0.f; acc1[c] = 0.f; accm[c] = 0.f; }
            for (int u0 = 0; u0 < n_units; u0 += UNR) {
                uint4 q[UNR], h[UNR];
#pragma unroll
                for (int i = 0; i < UNR; i++) { // all loads of the batch in flight first
                    const int un = u0 + i;
                    q[i] = make_uint4(0, 0, 0, 0);
                    h[i] = make_uint4(0, 0, 0, 0);
                    if (un < n_units) {
                        q[i] = ld_stream16(qg + (int64_t)un * unit_bytes);
                        if (WT == PS_Q4_K) h[i] = *(const uint4 *)(ag + (int64_t)un * aux_unit);
                        else { const uint2 t = *(const uint2 *)(ag + (int64_t)un * aux_unit); h[i].x = t.x; h[i].y = t.y; }
                    }
                }
#pragma unroll
                for (int i = 0; i < UNR; i++) {
                    if (u0 + i < n_units) {
#pragma unroll
                        for (int c = 0; c < BS; c++) {
                            const char *base = smem + c * p.col_bytes;
                            LAct a;
                            a.q32  = (const int *)base;
                            a.d    = (const float *)(base + Kp);
                            a.bs32 = (const int *)(a.d + nblk);
                            unit_dot<WT>(q[i], h[i], u0 + i, u, a, acc0[c], acc1[c], accm[c]);
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < BS; c++) yres[c][pass] = row_reduce<WT>(acc0[c], acc1[c], accm[c]);
        }
        // ---- epilogue: lane with u == 0 owns row grp*RG + r
        const GemvW &W    = p.w[wi];
        const int64_t row = grp * TR::RG + r;
        if (u == 0 && row < W.N) {
#pragma unroll
            for (int c = 0; c < BS; c++) {
                if (c < p.bs) {
                    float v;
                    if (EPI == 1) {
                        v = ps_silu_mul(yres[c][0], yres[c][EPI == 1 ? 1 : 0]);
                        p.w[0].out[c * p.w[0].ldo + row] = v;
                    } else {
                        v = yres[c][0];
                        if (W.bias) v = __fadd_rn(v, W.bias[row]);
                        if (p.residual && wi == 0) v = __fadd_rn(p.residual[c * W.ldo + row], v);
                        W.out[c * W.ldo + row] = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Decode kernel (one activation column).  A workgroup owns a row group; its NW waves split the units of the
// row (K) so that EVERY weight load of the group is in flight at once (the mat-vec is pure latency x bandwidth).
// Each wave turns its units into exact integer partials held in registers; the fp32 fma chains are then run
// in unit order by handing the 64 accumulators from wave to wave through LDS (NW short turns), which keeps
// the reference's summation order.  Workgroups are persistent over row groups, so the activation prologue
// (RMSNorm + quantization) is paid once per workgroup and overlaps the first group's weight loads.
template <int WT, int UPW> struct Part;
template <int UPW> struct Part<PS_Q4_K, UPW> { int s[UPW], pr[UPW]; float dd[UPW], dm[UPW]; };
template <int UPW> struct Part<PS_Q8_0, UPW> { int s[UPW][4]; float dd[UPW][4]; };
template <int UPW> struct Part<PS_Q4_0, UPW> { int sl[UPW][4], sh[UPW][4]; float dd[UPW][4]; };

template <int WT, int UPW>
__device__ __forceinline__ void unit_partials(const uint4 q, const uint4 h, const int unit, const int u, const LAct a,
                                              Part<WT, UPW> &P, const int i) {
    constexpr uint32_t M = 0x0F0F0F0Fu;
    if constexpr (WT == PS_Q4_K) {
        const uint32_t sc03 = h.y & 0x3f3f3f3fu;
        const uint32_t sc47 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
        const uint32_t mn03 = h.z & 0x3f3f3f3fu;
        const uint32_t mn47 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
        const int base = unit * 64 + u;
        const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
        int s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int yl = a.q32[base + j * 16], yh = a.q32[base + j * 16 + 8];
            const uint32_t scp = (j < 2) ? sc03 : sc47;
            s += bfe8(scp, (2 * j) & 3) * dot4((int)(wq[j] & M), yl, 0) + bfe8(scp, (2 * j + 1) & 3) * dot4((int)((wq[j] >> 4) & M), yh, 0);
        }
        const int v = u & 3;
        const uint32_t mp = (v < 2) ? mn03 : mn47;
        P.s[i]  = s;
        P.pr[i] = bfe8(mp, (2 * v) & 3) * a.bs32[unit * 8 + 2 * v] + bfe8(mp, (2 * v + 1) & 3) * a.bs32[unit * 8 + 2 * v + 1];
        const float yd = a.d[unit];
        P.dd[i] = __fmul_rn(yd, ps_h2f((uint16_t)(h.x & 0xffff)));
        P.dm[i] = __fmul_rn(-yd, ps_h2f((uint16_t)(h.x >> 16)));
    } else if constexpr (WT == PS_Q8_0) {
        const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
        const uint16_t dh[4] = {(uint16_t)(h.x & 0xffff), (uint16_t)(h.x >> 16), (uint16_t)(h.y & 0xffff), (uint16_t)(h.y >> 16)};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int blk = unit * 4 + b;
            P.s[i][b]  = dot4((int)wq[b], a.q32[blk * 8 + u], 0);
            P.dd[i][b] = __fmul_rn(ps_h2f(dh[b]), a.d[blk]);
        }
    } else {
        const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
        const uint16_t dh[4] = {(uint16_t)(h.x & 0xffff), (uint16_t)(h.x >> 16), (uint16_t)(h.y & 0xffff), (uint16_t)(h.y >> 16)};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int blk = unit * 4 + b;
            const int yl = a.q32[blk * 8 + u], yh = a.q32[blk * 8 + 4 + u];
            P.sl[i][b] = dot4((int)(wq[b] & M), yl, 0) - 8 * dot4(0x01010101, yl, 0);
            P.sh[i][b] = dot4((int)((wq[b] >> 4) & M), yh, 0) - 8 * dot4(0x01010101, yh, 0);
            P.dd[i][b] = __fmul_rn(ps_h2f(dh[b]), a.d[blk]);
        }
    }
}

template <int WT, int UPW>
__device__ __forceinline__ void unit_chain(const Part<WT, UPW> &P, const int i, float &acc0, float &acc1, float &accm) {
    if constexpr (WT == PS_Q4_K) {
        acc0 = __fmaf_rn(P.dd[i], (float)P.s[i], acc0);
        accm = __fmaf_rn(P.dm[i], (float)P.pr[i], accm);
    } else if constexpr (WT == PS_Q8_0) {
#pragma unroll
        for (int b = 0; b < 4; b++) acc0 = __fmaf_rn(P.dd[i][b], (float)P.s[i][b], acc0);
    } else {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            acc0 = __fmaf_rn(P.dd[i][b], (float)P.sl[i][b], acc0);
            acc1 = __fmaf_rn(P.dd[i][b], (float)P.sh[i][b], acc1);
        }
    }
}

template <int WT, int UPW, int NW, int EPI, int PRO>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 3 : (NW == 8 && EPI == 1 ? 2 : 4))) void gemv1_kernel(const GemvParams p) {
    using TR = WTraits<WT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ float hand[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t K = p.K;
    const int Kp   = (int)((K + TR::UNIT - 1) / TR::UNIT * TR::UNIT);
    const int nblk = Kp / TR::BLK, nb32 = Kp / 32, nb16 = Kp / 16;
    const int nblk_k = (int)(K / TR::BLK), nb16_k = (int)(K / 16);
    int8_t *lq   = (int8_t *)smem;
    float *ld    = (float *)(smem + Kp);
    int *lb      = (int *)(ld + nblk);
    int16_t *l16 = (int16_t *)(lb + nb32);
    LAct A;
    A.q32 = (const int *)lq; A.d = ld; A.bs32 = lb;

    const int r = (WT == PS_Q4_0) ? (lane >> 2) : (lane >> 3);
    const int u = (WT == PS_Q4_0) ? (lane & 3) : (lane & 7);
    const int n_units = Kp / TR::UNIT;
    const int tot     = (EPI == 1) ? 2 * n_units : n_units;   // EPI 1: gate units then up units
    const int upw     = (tot + NW - 1) / NW;                  // host guarantees upw <= UPW
    const int un0     = wave * upw;
    const int64_t aux_unit = (int64_t)TR::RG * (WT == PS_Q4_K ? 16 : 8);
    const int64_t n_tasks  = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    bool first = true;

    uint4 q[UPW], h[UPW];
    auto issue_loads = [&](int64_t task) { // every load of this wave's share of row group `task`
        int wi = 0;
        int64_t grp = task;
        if (EPI == 0) {
            if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
            if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
        }
#pragma unroll
        for (int i = 0; i < UPW; i++) {
            const int un = un0 + i;
            q[i] = make_uint4(0, 0, 0, 0);
            h[i] = make_uint4(0, 0, 0, 0);
            if (i < upw && un < tot) {
                const int wsel    = (EPI == 1) ? (un >= n_units ? 1 : 0) : wi;
                const int ul      = (EPI == 1 && un >= n_units) ? un - n_units : un;
                const GemvW &W    = p.w[wsel];
                const uint8_t *qg = W.qs + (grp * n_units + ul) * 1024 + (int64_t)lane * 16;
                const uint8_t *ag = W.aux + (grp * n_units + ul) * aux_unit + (int64_t)r * (WT == PS_Q4_K ? 16 : 8);
                q[i] = ld_stream16(qg);
                if (WT == PS_Q4_K) h[i] = *(const uint4 *)ag;
                else { const uint2 t = *(const uint2 *)ag; h[i].x = t.x; h[i].y = t.y; }
            }
        }
    };
    // tiles per wave follow from units per wave (a tile is 256 elements)
    constexpr int TPW = (UPW * TR::UNIT / 256 / (EPI == 1 ? 2 : 1)) > 0 ? (UPW * TR::UNIT / 256 / (EPI == 1 ? 2 : 1)) : 1;
    float4 xv[TPW], wv[TPW];
    // the activation row is requested FIRST, the first row group's weights right behind it: vmcnt retires in
    // order, so the prologue below only waits for the (L2-resident) activation while the weights stream in
    if (PRO != 0) ps_qrow_load<(PRO == 1 ? 1 : 0), TPW>(p.x, p.nw, K, xv, wv);
    if ((int64_t)blockIdx.x < n_tasks) issue_loads(blockIdx.x);

    for (int64_t task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        int wi = 0;
        int64_t grp = task;
        if (EPI == 0) {
            if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
            if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
        }
        if (first) { // activation -> LDS once per workgroup, overlapping the loads above
            first = false;
            if (Kp != K) {
                for (int i = (int)K + threadIdx.x * 4; i < Kp; i += NW * 64 * 4) *(int *)(lq + i) = 0;
                for (int i = nblk_k + threadIdx.x; i < nblk; i += NW * 64) ld[i] = 0.f;
                for (int i = nb16_k + threadIdx.x; i < nb16; i += NW * 64) l16[i] = 0;
            }
            if (PRO == 0) {
                for (int64_t i = threadIdx.x * 16; i < K; i += NW * 64 * 16) *(int4 *)(lq + i) = *(const int4 *)(p.aq + i);
                for (int i = threadIdx.x; i < nblk_k; i += NW * 64) ld[i] = p.ad[i];
                for (int i = threadIdx.x; i < nb16_k; i += NW * 64) l16[i] = p.abs16[i];
                __syncthreads();
            } else {
                ps_qrow_compute<TR::VDT, (PRO == 1 ? 1 : 0), TPW>(xv, wv, p.eps, K, lq, ld, l16, red);
            }
            for (int i = threadIdx.x; i < nb32; i += NW * 64) lb[i] = (int)l16[2 * i] + (int)l16[2 * i + 1];
            __syncthreads();
        }
        // ---- exact integer partials, in registers
        Part<WT, UPW> P;
#pragma unroll
        for (int i = 0; i < UPW; i++) {
            const int un = un0 + i;
            if (i < upw && un < tot) unit_partials<WT, UPW>(q[i], h[i], (EPI == 1 && un >= n_units) ? un - n_units : un, u, A, P, i);
        }
        // q/h are dead now: put the next row group's weights in flight so they stream during the chain turns
        if (task + gridDim.x < n_tasks) issue_loads(task + gridDim.x);
        // ---- fp32 chains in unit order: wave 0 -> 1 -> ... -> NW-1
        float acc0 = 0.f, acc1 = 0.f, accm = 0.f, ygate = 0.f;
#pragma unroll
        for (int t = 0; t < NW; t++) {
            if (wave == t) {
                if (t > 0) { acc0 = hand[0][lane]; acc1 = hand[1][lane]; accm = hand[2][lane]; ygate = hand[3][lane]; }
#pragma unroll
                for (int i = 0; i < UPW; i++) {
                    const int un = un0 + i;
                    if (i < upw && un < tot) {
                        if (EPI == 1 && un == n_units) { // gate row finished: reduce it, restart the chains for the up row
                            ygate = row_reduce<WT>(acc0, acc1, accm);
                            acc0 = 0.f; acc1 = 0.f; accm = 0.f;
                        }
                        unit_chain<WT, UPW>(P, i, acc0, acc1, accm);
                    }
                }
                if (t < NW - 1) { hand[0][lane] = acc0; hand[1][lane] = acc1; hand[2][lane] = accm; hand[3][lane] = ygate; }
            }
            __syncthreads();
        }
        if (wave == NW - 1) {
            const float y     = row_reduce<WT>(acc0, acc1, accm);
            const GemvW &W    = p.w[wi];
            const int64_t row = grp * TR::RG + r;
            if (u == 0 && row < W.N) {
                if (EPI == 1) {
                    p.w[0].out[row] = ps_silu_mul(ygate, y);
                } else {
                    float v = y;
                    if (W.bias) v = __fadd_rn(v, W.bias[row]);
                    if (p.residual && wi == 0) v = __fadd_rn(p.residual[row], v);
                    W.out[row] = v;
                }
            }
        }
    }
}

template <int WT, int UPW, int NW, int EPI, int PRO>
void launch_g1(hipStream_t st, int n_cu, const GemvParams &p) {
    const int64_t n_tasks = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    const size_t smem     = (size_t)p.col_bytes;
    static bool attr_set = false;
    static int occ = 0;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)gemv1_kernel<WT, UPW, NW, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    if (occ == 0) { // resident workgroups per CU for this instantiation (registers / LDS), queried once
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)gemv1_kernel<WT, UPW, NW, EPI, PRO>, NW * 64, 24 * 1024) != hipSuccess || nb < 1) nb = 1;
        occ = nb > 4 ? 4 : nb;
    }
    int64_t grid      = n_tasks;
    const int64_t cap = (int64_t)n_cu * occ;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((gemv1_kernel<WT, UPW, NW, EPI, PRO>), dim3((unsigned)grid), dim3(NW * 64), smem, st, p);
}

template <int WT, int UPW, int NW>
void launch_g1_ep(hipStream_t st, int n_cu, const GemvParams &p, int epi, int pro) {
    if (epi == 1) {
        if (pro == 1) launch_g1<WT, UPW, NW, 1, 1>(st, n_cu, p); else launch_g1<WT, UPW, NW, 1, 0>(st, n_cu, p);
    } else {
        if (pro == 0) launch_g1<WT, UPW, NW, 0, 0>(st, n_cu, p);
        else if (pro == 1) launch_g1<WT, UPW, NW, 0, 1>(st, n_cu, p);
        else launch_g1<WT, UPW, NW, 0, 2>(st, n_cu, p);
    }
}

// returns false when the row is too long for the register-resident kernel (falls back to gemv_kernel)
template <int WT>
bool launch_g1_wt(hipStream_t st, int n_cu, const GemvParams &p, int epi, int pro) {
    const int unit = WTraits<WT>::UNIT;
    const int n_units = (int)((p.K + unit - 1) / unit), tot = epi == 1 ? 2 * n_units : n_units;
    if (epi == 1 && pro == 2) return false;
    if (tot <= 8) launch_g1_ep<WT, 2, 4>(st, n_cu, p, epi, pro);
    else if (tot <= 16) launch_g1_ep<WT, 4, 4>(st, n_cu, p, epi, pro);
    else if (tot <= 32) launch_g1_ep<WT, 4, 8>(st, n_cu, p, epi, pro);
    else if (tot <= 64) launch_g1_ep<WT, 4, 16>(st, n_cu, p, epi, pro);
    else return false;
    return true;
}

template <int WT, int BS, int EPI, int PRO>
void launch_one(hipStream_t st, int n_cu, const GemvParams &p) {
    const int64_t n_tasks = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    const size_t smem     = (size_t)p.col_bytes * BS;
    int64_t grid          = (n_tasks + 3) / 4;
    const int64_t cap     = (int64_t)n_cu * (smem > 40 * 1024 ? 2 : 4);
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    static bool attr_set = false;
    if (!attr_set && smem > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)gemv_kernel<WT, BS, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemv_kernel<WT, BS, EPI, PRO>), dim3((unsigned)grid), dim3(256), smem, st, p);
}

template <int WT, int BS>
int launch_epi(hipStream_t st, int n_cu, const GemvParams &p, int epi, int pro) {
    if (epi == 1) {
        if (pro == 0) launch_one<WT, BS, 1, 0>(st, n_cu, p);
        else if (pro == 1) launch_one<WT, BS, 1, 1>(st, n_cu, p);
        else launch_one<WT, BS, 1, 2>(st, n_cu, p);
    } else {
        if (pro == 0) launch_one<WT, BS, 0, 0>(st, n_cu, p);
        else if (pro == 1) launch_one<WT, BS, 0, 1>(st, n_cu, p);
        else launch_one<WT, BS, 0, 2>(st, n_cu, p);
    }
    return 0;
}

template <int WT>
int launch_wt(hipStream_t st, int n_cu, const GemvParams &p, int epi, int pro) {
    if (p.bs == 1 && launch_g1_wt<WT>(st, n_cu, p, epi, pro)) return 0;
    if (p.bs == 1) return launch_epi<WT, 1>(st, n_cu, p, epi, pro);
    if (p.bs <= 4) return launch_epi<WT, 4>(st, n_cu, p, epi, pro);
    return 3;
}

} // namespace

size_t psk_gemv_lds_col_bytes(int wt, int64_t K) {
    const int64_t blk = (wt == PS_Q4_K) ? 256 : 32, unit = (wt == PS_Q4_K) ? 256 : 128;
    const int64_t Kp = (K + unit - 1) / unit * unit;
    const size_t b = (size_t)Kp + (size_t)(Kp / blk) * 4 + (size_t)(Kp / 32) * 4 + (size_t)(Kp / 16) * 2;
    return (b + 15) / 16 * 16;
}

// bs <= 4 columns per launch; larger batches are split by the caller.
int psk_gemv(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int vdt, int64_t K, int64_t bs) {
    GemvParams p{};
    p.n_w          = a.n_w;
    p.groups_total = 0;
    p.K            = K;
    p.bs           = bs;
    p.residual     = a.residual;
    p.x            = a.pro_x;
    p.nw           = a.pro_norm_w;
    p.eps          = a.pro_eps;
    p.aq           = act.qs;
    p.ad           = act.d;
    p.abs16        = act.bs16;
    const int wt   = a.w[0]->dtype;
    const int rg   = (wt == PS_Q4_0) ? 16 : 8;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != wt || a.w[i]->K != K) return 4;
        const int64_t ng = (a.w[i]->N + rg - 1) / rg;
        p.w[i] = GemvW{a.w[i]->qs, a.w[i]->aux, a.out[i], a.bias[i], a.w[i]->N, a.ldo[i], ng};
        p.groups_total += ng;
    }
    (void)vdt;
    p.col_bytes = (int64_t)psk_gemv_lds_col_bytes(wt, K);
    if ((size_t)p.col_bytes * (bs == 1 ? 1 : 4) > 150 * 1024) return 7;
    const int epi = a.silu_pair ? 1 : 0;
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N)) return 5;
    switch (wt) {
    case PS_Q4_0: return launch_wt<PS_Q4_0>(st, n_cu, p, epi, a.pro);
    case PS_Q8_0: return launch_wt<PS_Q8_0>(st, n_cu, p, epi, a.pro);
    case PS_Q4_K: return launch_wt<PS_Q4_K>(st, n_cu, p, epi, a.pro);
    }
    return 6;
}
