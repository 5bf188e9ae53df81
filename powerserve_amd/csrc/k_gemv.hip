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
#include "ps_gemv_dev.h"

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
    unsigned long long *dbg; // timeline buffer or null (ps_hip_debug_timeline)
    int split_q, split_r;    // (unused by the kernels of this file since gemv3 went)
    psk_rope_kv rope;        // EPI 2
};

// EPI 0: out = y (+bias) (+residual).   EPI 1: out[0] = silu(y_w0) * y_w1 (same row of w[0] and w[1]).
// NWV waves per workgroup share one LDS copy of the BS activation columns; a wave owns a row group at a time and
// keeps the fma chains of all BS columns in its own registers (no exchange).  BS 1/4 with 4 waves: small batches;
// BS 8/16 with 16 waves: prefill / tree-verify column groups from pre-quantized activations (PRO 0).
template <int WT, int BS, int EPI, int PRO, int NWV>
__global__ __launch_bounds__(NWV * 64) void gemv_kernel(const GemvParams p) {
    constexpr int NT = NWV * 64;
    using TR = WTraits<WT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
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
                for (int i = (int)K + threadIdx.x * 4; i < Kp; i += NT * 4) *(int *)(lq + i) = 0;
                for (int i = nblk_k + threadIdx.x; i < nblk; i += NT) ld[i] = 0.f;
                for (int i = nb16_k + threadIdx.x; i < nb16; i += NT) l16[i] = 0;
            }
            if (PRO == 0) {
                for (int64_t i = threadIdx.x * 16; i < K; i += NT * 16) *(int4 *)(lq + i) = *(const int4 *)(p.aq + col * K + i);
                for (int i = threadIdx.x; i < nblk_k; i += NT) ld[i] = p.ad[col * nblk_k + i];
                for (int i = threadIdx.x; i < nb16_k; i += NT) l16[i] = p.abs16[col * nb16_k + i];
                __syncthreads();
            } else {
                ps_quantize_row_wg<TR::VDT, PRO == 1 ? 1 : 0, 64 / NWV>(p.x + col * K, p.nw, p.eps, K, lq, ld, l16, red);
            }
            for (int i = threadIdx.x; i < nb32; i += NT) lb[i] = (int)l16[2 * i] + (int)l16[2 * i + 1];
        } else {
            for (int i = threadIdx.x * 4; i < Kp; i += NT * 4) *(int *)(lq + i) = 0;
            for (int i = threadIdx.x; i < nblk; i += NT) ld[i] = 0.f;
            for (int i = threadIdx.x; i < nb32; i += NT) lb[i] = 0;
        }
    }
    __syncthreads();

    const int r = (WT == PS_Q4_0) ? (lane >> 2) : (lane >> 3);
    const int u = (WT == PS_Q4_0) ? (lane & 3) : (lane & 7);
    const int n_units        = (int)((K + TR::UNIT - 1) / TR::UNIT);
    const int64_t unit_bytes = 1024;                                   // quant-plane bytes per unit per row group
    const int64_t aux_unit   = (int64_t)TR::RG * (WT == PS_Q4_K ? 16 : 8);
    const int64_t n_tasks    = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    constexpr int UNR = BS >= 8 ? 2 : 4; // weight units in flight per wave (the column accumulators need the registers)

    for (int64_t task = (int64_t)blockIdx.x * NWV + wave; task < n_tasks; task += (int64_t)gridDim.x * NWV) {
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
    static unsigned long long attr_set = 0; // devices that have the attribute
    static int occ = 0;
    if (ps_first_on_device(&attr_set)) {
        (void)hipFuncSetAttribute((const void *)gemv1_kernel<WT, UPW, NW, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
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
    psk_note_kernel("gemv1_kernel<%d, %d, %d, %d, %d>", WT, UPW, NW, EPI, PRO);
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

// ---------------------------------------------------------------------------------------------------------
// (The first producer / consumer decode kernel, gemv3 -- 14 producer waves + 2 chain waves per 1024-thread workgroup, integer
//  records, the chain waves deriving every block scale -- lived here through round 2.  Every shape it took now goes to its
//  successors: Q4_K to gemv4 (k_gemv4.hip), Q4_0 / Q8_0 to gemvb (k_gemvb.hip), Q6_K / Q5_K to gemvk (k_gemvk.hip); what they do
//  not cover -- row lengths that are not whole units of four, very short rows, the long single-matrix stream of a Q4_0 / Q8_0
//  lm_head -- was never gemv3's either and takes gemv1 / gemv_kernel below.  Deleted in round 3; DESIGN.md 5 keeps its history.)
constexpr int G3_DBG_WGS = 1024; // workgroups with a timeline slot (ps_hip_debug_timeline: gemv4, gemm4k, the attention kernels)
unsigned long long *g_dbg_buf = nullptr;
int g_dbg_key = -1;

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

template <int WT, int BS, int EPI, int PRO, int NWV>
void launch_one(hipStream_t st, int n_cu, const GemvParams &p) {
    const int64_t n_tasks = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    const size_t smem     = (size_t)p.col_bytes * BS;
    int64_t grid          = (n_tasks + NWV - 1) / NWV;
    const int64_t cap     = (int64_t)n_cu * (NWV == 16 ? 1 : (smem > 40 * 1024 ? 2 : 4));
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    static unsigned long long attr_set = 0; // devices that have the attribute
    if (ps_first_on_device(&attr_set) && smem > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)gemv_kernel<WT, BS, EPI, PRO, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    }
    psk_note_kernel("gemv_kernel<%d, %d, %d, %d, %d>", WT, BS, EPI, PRO, NWV);
    hipLaunchKernelGGL((gemv_kernel<WT, BS, EPI, PRO, NWV>), dim3((unsigned)grid), dim3(NWV * 64), smem, st, p);
}

template <int WT, int BS>
int launch_epi(hipStream_t st, int n_cu, const GemvParams &p, int epi, int pro) {
    if (epi == 1) {
        if (pro == 0) launch_one<WT, BS, 1, 0, 4>(st, n_cu, p);
        else if (pro == 1) launch_one<WT, BS, 1, 1, 4>(st, n_cu, p);
        else launch_one<WT, BS, 1, 2, 4>(st, n_cu, p);
    } else {
        if (pro == 0) launch_one<WT, BS, 0, 0, 4>(st, n_cu, p);
        else if (pro == 1) launch_one<WT, BS, 0, 1, 4>(st, n_cu, p);
        else launch_one<WT, BS, 0, 2, 4>(st, n_cu, p);
    }
    return 0;
}

template <int WT>
int launch_wt(hipStream_t st, int n_cu, const GemvParams &p, int epi, int pro) {
    if (epi == 2) return 8; // the fused RoPE epilogue exists in the producer / chain-wave kernels only (psk_gemv_rope_ok)
    if (p.bs == 1 && launch_g1_wt<WT>(st, n_cu, p, epi, pro)) return 0;
    if (p.bs == 1) return launch_epi<WT, 1>(st, n_cu, p, epi, pro);
    if (p.bs <= 4) return launch_epi<WT, 4>(st, n_cu, p, epi, pro);
    return 3;
}

// Activation image of 8 / 16 pre-quantized columns in LDS, shared by gemm8m_kernel and gemm8b_kernel.
template <int WT, int NT, int C = 8>
__device__ __forceinline__ void gemm8_stage(const int8_t *aq, const float *ad, const int16_t *abs16, const int K, char *smem, const int c0,
                                            const int nc) {
    using TR = WTraits<WT>;
    const int nblk = K / TR::BLK;
    // LDS image: one record per (unit, column), records of a unit adjacent, so that every per-column read of the inner
    // loop is  base(unit, lane) + compile-time offset:
    //   Q4_K  REC 304: [256 B quants, dwords transposed (g, u) -> [u][g]] [8 int sums of 32] [float d] [pad]
    //   Q8_0 / Q4_0  REC 144: [128 B quants, transposed] [4 float d]
    constexpr int REC = (WT == PS_Q4_K) ? 304 : 144, UDW = TR::UNIT / 4; // dwords of quants per unit and column
    auto tpos = [](int i) { // dword i of a unit -> its place in the transposed record
        if (WT == PS_Q4_K) return (i & 7) * 8 + (i >> 3);                 // (g, u) -> [u][g]
        if (WT == PS_Q8_0) return (i & 7) * 4 + (i >> 3);                 // (block b, d) -> [d][b]
        return (i & 3) * 8 + (i >> 3) * 2 + ((i >> 2) & 1);               // Q4_0: (b, half*4 + u') -> [u'][b][half]
    };
    {   // 16-byte global loads, SB of them in flight per thread, then the transposed dword stores
        constexpr int SB = 8;
        const int n16 = C * (K / 16); // int4 pieces of the 8 columns (K % 128 == 0)
        for (int base = threadIdx.x; base < n16; base += NT * SB) {
            int4 v[SB];
#pragma unroll
            for (int k = 0; k < SB; k++) {
                const int idx = base + k * NT, c = idx / (K / 16), i = idx % (K / 16);
                v[k] = (idx < n16 && c < nc) ? ((const int4 *)(aq + (int64_t)(c0 + c) * K))[i] : make_int4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < SB; k++) {
                const int idx = base + k * NT, c = idx / (K / 16), i = (idx % (K / 16)) * 4;
                if (idx < n16) {
                    int *rec = (int *)(smem + ((i / UDW) * C + c) * REC);
                    const int w = i % UDW; // (four consecutive dwords never straddle a unit)
                    rec[tpos(w)] = v[k].x; rec[tpos(w + 1)] = v[k].y; rec[tpos(w + 2)] = v[k].z; rec[tpos(w + 3)] = v[k].w;
                }
            }
        }
    }
    constexpr int BPU = TR::UNIT / TR::BLK; // scales per unit (1 or 4)
    for (int idx = threadIdx.x; idx < C * nblk; idx += NT) {
        const int c = idx / nblk, i = idx % nblk;
        *(float *)(smem + ((i / BPU) * C + c) * REC + (WT == PS_Q4_K ? 288 : 128) + (i % BPU) * 4) = c < nc ? ad[(int64_t)(c0 + c) * nblk + i] : 0.f;
    }
    if (WT == PS_Q4_K) {
        for (int idx = threadIdx.x; idx < C * (K / 32); idx += NT) {
            const int c = idx / (K / 32), i = idx % (K / 32);
            const int16_t *b = abs16 + (int64_t)(c0 + c) * (K / 16) + 2 * i;
            *(int *)(smem + ((i / 8) * C + c) * REC + 256 + (i % 8) * 4) = c < nc ? (int)b[0] + (int)b[1] : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Batched mat-mul (prefill chunks, tree verify) for the shapes the fp16 matrix-core kernels of k_gemm4k.hip do not take: 8 or 16
// activation columns per workgroup.  grid = (row-group tiles, column groups); a workgroup stages its pre-quantized columns in LDS
// once (gemm8_stage: quants transposed so that the bytes a lane needs per unit are contiguous, ds_read_b128), every wave then owns
// one row group: the weights of a unit are unpacked ONCE and meet the columns, each column keeping the reference's fma chains in
// this lane's registers.  Same arithmetic, same order as the mat-vec.  (Round 1's v_dot4 form, gemm8_kernel, was deleted in round
// 3: the two kernels below -- the same lane roles with the quad dots on v_mfma_i32_4x4x4_16B_i8 -- took every launch.)

// ---------------------------------------------------------------------------------------------------------
// Q4_K batched mat-mul with the quad dots on the matrix cores (exact: everything before the fp32 chain is integer).
// v_mfma_i32_4x4x4_16B_i8 computes 16 independent 4 x 4 products with K = 4 — a "quad" of the reference (the four
// elements one AVX lane u owns in a 32-element sub-block).  Block b = lane >> 2; result lane l, register r holds
// dot4(A operand of lane 4b + r, B operand of lane l) (tools/micro/mfma4probe.hip).  Lane roles:
//   qs = lane >> 4 (owns u = qs and u = qs + 4), rq = (lane >> 3) & 1, cq = (lane >> 2) & 1, s = lane & 3
//   A operand: weight row rq*4 + s of the row group;  B operand and results: column cq*4 + s;  registers: rows rq*4 + r.
// The 6-bit sub-block scale is folded into the A operand as two 3-bit factors (quad nibbles * factor <= 105 stay bytes:
// one v_pk_mul_lo_u16), so the eight quads of a 256-element unit accumulate in the MFMA and
//   sumi[u] = 8 * S_hi + S_lo = sum_j scale_j * dot_j  exactly;  the fp32 chain (d * sumi, dmin * mins.bsums) is the
// reference's, 4 rows x 2 u per lane.  Per unit and lane: 32 MFMAs and ~130 VALU instructions instead of ~220.
typedef int ps_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short ps_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pk_mul_u16(uint32_t a, uint32_t b) { // two independent 16-bit products
    ps_u16x2 va, vb;
    __builtin_memcpy(&va, &a, 4);
    __builtin_memcpy(&vb, &b, 4);
    const ps_u16x2 vr = va * vb;
    int r;
    __builtin_memcpy(&r, &vr, 4);
    return r;
}
template <int EPI, int NWV, int C>
__global__ __launch_bounds__(NWV * 64) void gemm8m_kernel(const GemvParams p) {
    constexpr int NT = NWV * 64, REC = 304, NCS = C / 8; // NCS column sets of 8: lane's columns are bcol + 8 cs
    constexpr uint32_t M = 0x0F0F0F0Fu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = (int)p.K, n_units = K / 256, c0 = blockIdx.y * C, nc = min(C, (int)p.bs - c0);
    gemm8_stage<PS_Q4_K, NT, C>(p.aq, p.ad, p.abs16, K, smem, c0, nc);
    __syncthreads();

    const int qs = lane >> 4, rq = (lane >> 3) & 1, cq = (lane >> 2) & 1, s4 = lane & 3;
    const int arow = rq * 4 + s4, bcol = cq * 4 + s4;
    const int64_t n_tasks = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    const int64_t task = (int64_t)blockIdx.x * NWV + wave;
    if (task >= n_tasks) return;
    float yg[NCS][4];
#pragma unroll
    for (int pass = 0; pass < (EPI == 1 ? 2 : 1); pass++) {
        int wi = 0;
        int64_t grp = task;
        if (EPI == 1) {
            wi = pass;
        } else {
            if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
            if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
        }
        const uint8_t *qsb = p.w[0].qs, *ax = p.w[0].aux;
        if (wi == 1) { qsb = p.w[1].qs; ax = p.w[1].aux; }
        if (wi == 2) { qsb = p.w[2].qs; ax = p.w[2].aux; }
        const uint8_t *qg = qsb + grp * n_units * 1024 + (arow * 8 + qs) * 16; // piece (row, u = qs); (row, qs + 4) sits 64 bytes on
        const uint8_t *ag = ax + grp * n_units * 128 + arow * 16;
        float acc0[NCS][2][4], accm[NCS][4];
#pragma unroll
        for (int cs = 0; cs < NCS; cs++)
#pragma unroll
            for (int r = 0; r < 4; r++) { acc0[cs][0][r] = 0.f; acc0[cs][1][r] = 0.f; accm[cs][r] = 0.f; }

        auto unit = [&](const int un, const uint4 qa, const uint4 qb, const uint4 h) {
            // ---- this lane's weight row: scales as two 3-bit factors, replicated into both 16-bit halves
            const uint32_t sc03 = h.y & 0x3f3f3f3fu, sc47 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
            const uint32_t mn03 = h.z & 0x3f3f3f3fu, mn47 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
            const uint32_t lo[2] = {sc03 & 0x07070707u, sc47 & 0x07070707u}, hi[2] = {(sc03 >> 3) & 0x07070707u, (sc47 >> 3) & 0x07070707u};
            const uint32_t mp = (qs < 2) ? mn03 : mn47;
            const int mnp = (int)(bfe8(mp, (2 * qs) & 3) | (bfe8(mp, (2 * qs + 1) & 3) << 16)); // {min[2 qs], min[2 qs + 1]} as int16 pair
            const float dwf = ps_h2f((uint16_t)(h.x & 0xffff)), dmf = ps_h2f((uint16_t)(h.x >> 16));
            // ---- B operands: this lane's column of every set, quants of u = qs and u = qs + 4 (dword 2 jj + half = quad (jj, half))
            const uint32_t wa[4] = {qa.x, qa.y, qa.z, qa.w}, wb[4] = {qb.x, qb.y, qb.z, qb.w};
            int ya[NCS][8], yb[NCS][8];
            ps_i32x4 slo[NCS][2], shi[NCS][2];
#pragma unroll
            for (int cs = 0; cs < NCS; cs++) {
                const char *rec = smem + (un * C + bcol + 8 * cs) * REC;
                const int4 ya0 = *(const int4 *)(rec + qs * 32), ya1 = *(const int4 *)(rec + qs * 32 + 16);
                const int4 yb0 = *(const int4 *)(rec + (qs + 4) * 32), yb1 = *(const int4 *)(rec + (qs + 4) * 32 + 16);
                ya[cs][0] = ya0.x; ya[cs][1] = ya0.y; ya[cs][2] = ya0.z; ya[cs][3] = ya0.w; ya[cs][4] = ya1.x; ya[cs][5] = ya1.y; ya[cs][6] = ya1.z; ya[cs][7] = ya1.w;
                yb[cs][0] = yb0.x; yb[cs][1] = yb0.y; yb[cs][2] = yb0.z; yb[cs][3] = yb0.w; yb[cs][4] = yb1.x; yb[cs][5] = yb1.y; yb[cs][6] = yb1.z; yb[cs][7] = yb1.w;
                slo[cs][0] = slo[cs][1] = shi[cs][0] = shi[cs][1] = ps_i32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int sbi = 0; sbi < 8; sbi++) { // sub-block 2 jj + half: low / high nibbles of dword jj; A operands once for all sets
                const int jj = sbi >> 1, half = sbi & 1;
                const uint32_t sel = 0x0c000c00u | (uint32_t)(sbi & 3) | ((uint32_t)(sbi & 3) << 16); // byte sbi & 3 -> both halves
                const uint32_t flo = __builtin_amdgcn_perm(0u, lo[sbi >> 2], sel), fhi = __builtin_amdgcn_perm(0u, hi[sbi >> 2], sel);
                const uint32_t na = (half ? wa[jj] >> 4 : wa[jj]) & M, nb = (half ? wb[jj] >> 4 : wb[jj]) & M;
                const int alo0 = pk_mul_u16(na, flo), ahi0 = pk_mul_u16(na, fhi), alo1 = pk_mul_u16(nb, flo), ahi1 = pk_mul_u16(nb, fhi);
#pragma unroll
                for (int cs = 0; cs < NCS; cs++) {
                    slo[cs][0] = __builtin_amdgcn_mfma_i32_4x4x4i8(alo0, ya[cs][sbi], slo[cs][0], 0, 0, 0);
                    shi[cs][0] = __builtin_amdgcn_mfma_i32_4x4x4i8(ahi0, ya[cs][sbi], shi[cs][0], 0, 0, 0);
                    slo[cs][1] = __builtin_amdgcn_mfma_i32_4x4x4i8(alo1, yb[cs][sbi], slo[cs][1], 0, 0, 0);
                    shi[cs][1] = __builtin_amdgcn_mfma_i32_4x4x4i8(ahi1, yb[cs][sbi], shi[cs][1], 0, 0, 0);
                }
            }
            // ---- fp32 chains of rows rq*4 + r: the row's d, dmin, mins come from lane r of this quad
#pragma unroll
            for (int cs = 0; cs < NCS; cs++) {
                const char *rec = smem + (un * C + bcol + 8 * cs) * REC;
                const int2 bs = *(const int2 *)(rec + 256 + qs * 8);
                const float yd = *(const float *)(rec + 288);
                const int bsp = (bs.x & 0xffff) | (bs.y << 16); // (|bsums of 32| <= 4064)
#define PS_G8M_ROW(r, CTRL)                                                                                     \
                {                                                                                               \
                    const float dwr = dpp_f<CTRL>(dwf), dmr = dpp_f<CTRL>(dmf);                                 \
                    const int mnr   = dpp_i<CTRL>(mnp);                                                         \
                    const float d = __fmul_rn(yd, dwr), dmin = __fmul_rn(-yd, dmr);                             \
                    acc0[cs][0][r] = __fmaf_rn(d, (float)(shi[cs][0][r] * 8 + slo[cs][0][r]), acc0[cs][0][r]);  \
                    acc0[cs][1][r] = __fmaf_rn(d, (float)(shi[cs][1][r] * 8 + slo[cs][1][r]), acc0[cs][1][r]);  \
                    accm[cs][r]    = __fmaf_rn(dmin, (float)dot2_i16((uint32_t)mnr, (uint32_t)bsp, 0), accm[cs][r]); \
                }
                PS_G8M_ROW(0, 0x00) PS_G8M_ROW(1, 0x55) PS_G8M_ROW(2, 0xAA) PS_G8M_ROW(3, 0xFF)
#undef PS_G8M_ROW
            }
        };
        auto load_h = [&](int un) { return *(const uint4 *)(ag + (int64_t)un * 128); };
        {   // PF units in flight per wave (register ring, no copies); loads unconditional (index clamped to the last unit)
            constexpr int PF = (NWV == 16) ? 2 : 4;
            uint4 qar[PF], qbr[PF], hr[PF];
#pragma unroll
            for (int s = 0; s < PF; s++) { const int uc = min(s, n_units - 1); qar[s] = ld_stream16(qg + (int64_t)uc * 1024); qbr[s] = ld_stream16(qg + (int64_t)uc * 1024 + 64); hr[s] = load_h(uc); }
            for (int un0 = 0; un0 < n_units; un0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; s++) {
                    const int un = un0 + s;
                    if (un >= n_units) break;
                    const uint4 qa = qar[s], qb = qbr[s], h = hr[s];
                    const int nx = min(un + PF, n_units - 1);
                    qar[s] = ld_stream16(qg + (int64_t)nx * 1024); qbr[s] = ld_stream16(qg + (int64_t)nx * 1024 + 64); hr[s] = load_h(nx);
                    unit(un, qa, qb, h);
                }
            }
        }
        // ---- epilogue: hsum_float_8 over u (u and u + 4 are this lane's; u +- 2, u +- 1 sit 32 and 16 lanes away) + acc_m
        int64_t Nw = p.w[0].N, ldo = p.w[0].ldo;
        float *o = p.w[0].out;
        const float *b = p.w[0].bias;
        if (wi == 1) { Nw = p.w[1].N; ldo = p.w[1].ldo; o = p.w[1].out; b = p.w[1].bias; }
        if (wi == 2) { Nw = p.w[2].N; ldo = p.w[2].ldo; o = p.w[2].out; b = p.w[2].bias; }
#pragma unroll
        for (int cs = 0; cs < NCS; cs++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = __fadd_rn(acc0[cs][0][r], acc0[cs][1][r]);
            v = __fadd_rn(v, __shfl_xor(v, 32, 64));
            v = __fadd_rn(v, __shfl_xor(v, 16, 64));
            float mm = __fadd_rn(accm[cs][r], __shfl_xor(accm[cs][r], 32, 64));
            mm = __fadd_rn(mm, __shfl_xor(mm, 16, 64));
            const float y = __fadd_rn(v, mm);
            if (EPI == 1 && pass == 0) { yg[cs][r] = y; continue; }
            const int64_t row = grp * 8 + rq * 4 + r;
            const int col = bcol + 8 * cs;
            if (qs == 0 && row < Nw && col < nc) {
                if (EPI == 1) {
                    p.w[0].out[(int64_t)(c0 + col) * p.w[0].ldo + row] = ps_silu_mul(yg[cs][r], y);
                } else {
                    float val = y;
                    if (b) val = __fadd_rn(val, b[row]);
                    if (p.residual && wi == 0) val = __fadd_rn(p.residual[(int64_t)(c0 + col) * ldo + row], val);
                    o[(int64_t)(c0 + col) * ldo + row] = val;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Q8_0 / Q4_0 batched mat-mul with the quad dots on the matrix cores (same idea as gemm8m_kernel; here every quad dot
// meets its own fp32 block scale, so the MFMAs do not accumulate: D = dot4(weight quad, activation quad), then
// acc[u] = fma(d_w * d_y, (float)D, acc[u]) block after block as in ggml_vec_dot_q8_0_q8_0 / q4_0_q8_0).
//   Q8_0 (row groups of 8):  qs = lane >> 4 owns u = qs, qs + 4;  A row (lane >> 3 & 1) * 4 + s;  column (lane >> 2 & 1) * 4 + s
//   Q4_0 (row groups of 16): qs = lane >> 4: the low nibbles of piece (row, qs) are u = qs, the high ones u = qs + 4;
//                            A row (lane >> 2 & 3) * 4 + s;  columns s and s + 4 (two sets)
template <int WT, int EPI, int NWV>
__global__ __launch_bounds__(NWV * 64) void gemm8b_kernel(const GemvParams p) {
    using TR = WTraits<WT>;
    constexpr int NT = NWV * 64, C = 8, REC = 144, NCS = (WT == PS_Q4_0) ? 2 : 1;
    constexpr uint32_t M = 0x0F0F0F0Fu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = (int)p.K, n_units = K / 128, c0 = blockIdx.y * C, nc = min(C, (int)p.bs - c0);
    gemm8_stage<WT, NT, C>(p.aq, p.ad, p.abs16, K, smem, c0, nc);
    __syncthreads();

    const int qs = lane >> 4, s4 = lane & 3;
    const int rqi  = (WT == PS_Q4_0) ? (lane >> 2) & 3 : (lane >> 3) & 1; // row quad of the row group
    const int arow = rqi * 4 + s4;
    const int col0 = (WT == PS_Q4_0) ? s4 : ((lane >> 2) & 1) * 4 + s4;   // column of set cs: col0 + 4 cs (Q4_0 only has cs = 1)
    const int64_t n_tasks = (EPI == 1) ? p.w[0].n_groups : p.groups_total;
    const int64_t task = (int64_t)blockIdx.x * NWV + wave;
    if (task >= n_tasks) return;
    float yg[NCS][4];
#pragma unroll
    for (int pass = 0; pass < (EPI == 1 ? 2 : 1); pass++) {
        int wi = 0;
        int64_t grp = task;
        if (EPI == 1) {
            wi = pass;
        } else {
            if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
            if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
        }
        const uint8_t *qsb = p.w[0].qs, *ax = p.w[0].aux;
        if (wi == 1) { qsb = p.w[1].qs; ax = p.w[1].aux; }
        if (wi == 2) { qsb = p.w[2].qs; ax = p.w[2].aux; }
        // Q8_0: pieces (row, qs) and (row, qs + 4), 64 bytes apart;  Q4_0: the one piece (row, qs)
        const uint8_t *qg = qsb + grp * n_units * 1024 + ((WT == PS_Q4_0) ? (arow * 4 + qs) : (arow * 8 + qs)) * 16;
        const uint8_t *ag = ax + grp * n_units * (TR::RG * 8) + arow * 8;
        float acc[NCS][2][4];
#pragma unroll
        for (int cs = 0; cs < NCS; cs++)
#pragma unroll
            for (int r = 0; r < 4; r++) { acc[cs][0][r] = 0.f; acc[cs][1][r] = 0.f; }

        auto unit = [&](const int un, const uint4 qa, const uint4 qb, const uint2 h) {
            // A operands: [u half][block]
            int wa[2][4];
            const uint32_t w0[4] = {qa.x, qa.y, qa.z, qa.w}, w1[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (WT == PS_Q8_0) { wa[0][b] = (int)w0[b]; wa[1][b] = (int)w1[b]; }
                else { // nibble - 8 as signed bytes
                    wa[0][b] = (int)((((w0[b] & M) | 0x80808080u) - 0x08080808u) ^ 0x80808080u);
                    wa[1][b] = (int)(((((w0[b] >> 4) & M) | 0x80808080u) - 0x08080808u) ^ 0x80808080u);
                }
            }
            // this lane's row: the four fp16 block scales as floats
            const float dwo[4] = {ps_h2f((uint16_t)(h.x & 0xffff)), ps_h2f((uint16_t)(h.x >> 16)), ps_h2f((uint16_t)(h.y & 0xffff)), ps_h2f((uint16_t)(h.y >> 16))};
#pragma unroll
            for (int cs = 0; cs < NCS; cs++) {
                const char *rec = smem + (un * C + col0 + 4 * cs) * REC;
                int y[2][4]; // [u half][block]
                if (WT == PS_Q8_0) {
                    const int4 t0 = *(const int4 *)(rec + qs * 16), t1 = *(const int4 *)(rec + (qs + 4) * 16);
                    y[0][0] = t0.x; y[0][1] = t0.y; y[0][2] = t0.z; y[0][3] = t0.w;
                    y[1][0] = t1.x; y[1][1] = t1.y; y[1][2] = t1.z; y[1][3] = t1.w;
                } else { // record of u' = qs: dword 2 b + half
                    const int4 t0 = *(const int4 *)(rec + qs * 32), t1 = *(const int4 *)(rec + qs * 32 + 16);
                    y[0][0] = t0.x; y[1][0] = t0.y; y[0][1] = t0.z; y[1][1] = t0.w;
                    y[0][2] = t1.x; y[1][2] = t1.y; y[0][3] = t1.z; y[1][3] = t1.w;
                }
                const float4 yd = *(const float4 *)(rec + 128);
                const float ydv[4] = {yd.x, yd.y, yd.z, yd.w};
#define PS_G8B_ROW(r, CTRL)                                                                                  \
                    {                                                                                        \
                        const float d = __fmul_rn(dpp_f<CTRL>(dwo[b]), ydv[b]);                               \
                        acc[cs][0][r] = __fmaf_rn(d, (float)D0[r], acc[cs][0][r]);                            \
                        acc[cs][1][r] = __fmaf_rn(d, (float)D1[r], acc[cs][1][r]);                            \
                    }
#pragma unroll
                for (int b = 0; b < 4; b++) { // block after block: the products of a block are consumed before the next ones exist
                    const ps_i32x4 D0 = __builtin_amdgcn_mfma_i32_4x4x4i8(wa[0][b], y[0][b], ps_i32x4{0, 0, 0, 0}, 0, 0, 0);
                    const ps_i32x4 D1 = __builtin_amdgcn_mfma_i32_4x4x4i8(wa[1][b], y[1][b], ps_i32x4{0, 0, 0, 0}, 0, 0, 0);
                    PS_G8B_ROW(0, 0x00) PS_G8B_ROW(1, 0x55) PS_G8B_ROW(2, 0xAA) PS_G8B_ROW(3, 0xFF)
                }
#undef PS_G8B_ROW
            }
        };
        auto load_h = [&](int un) { return *(const uint2 *)(ag + (int64_t)un * (TR::RG * 8)); };
        {   // PF units in flight per wave (register ring); loads unconditional (index clamped to the last unit)
            constexpr int PF = (NWV == 4) ? 4 : 2;
            uint4 qar[PF], qbr[PF];
            uint2 hr[PF];
#pragma unroll
            for (int s = 0; s < PF; s++) {
                const int uc = min(s, n_units - 1);
                qar[s] = ld_stream16(qg + (int64_t)uc * 1024);
                qbr[s] = (WT == PS_Q8_0) ? ld_stream16(qg + (int64_t)uc * 1024 + 64) : make_uint4(0, 0, 0, 0);
                hr[s] = load_h(uc);
            }
            for (int un0 = 0; un0 < n_units; un0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; s++) {
                    const int un = un0 + s;
                    if (un >= n_units) break;
                    const uint4 qa = qar[s], qb = qbr[s];
                    const uint2 h = hr[s];
                    const int nx = min(un + PF, n_units - 1);
                    qar[s] = ld_stream16(qg + (int64_t)nx * 1024);
                    if (WT == PS_Q8_0) qbr[s] = ld_stream16(qg + (int64_t)nx * 1024 + 64);
                    hr[s] = load_h(nx);
                    unit(un, qa, qb, h);
                }
            }
        }
        // ---- epilogue: hsum_float_8 over u (u and u + 4 are this lane's; u +- 2, u +- 1 sit 32 and 16 lanes away)
        int64_t Nw = p.w[0].N, ldo = p.w[0].ldo;
        float *o = p.w[0].out;
        const float *b = p.w[0].bias;
        if (wi == 1) { Nw = p.w[1].N; ldo = p.w[1].ldo; o = p.w[1].out; b = p.w[1].bias; }
        if (wi == 2) { Nw = p.w[2].N; ldo = p.w[2].ldo; o = p.w[2].out; b = p.w[2].bias; }
#pragma unroll
        for (int cs = 0; cs < NCS; cs++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = __fadd_rn(acc[cs][1][r], acc[cs][0][r]); // a[k + 4] + a[k]
            v = __fadd_rn(v, __shfl_xor(v, 32, 64));
            v = __fadd_rn(v, __shfl_xor(v, 16, 64));
            if (EPI == 1 && pass == 0) { yg[cs][r] = v; continue; }
            const int64_t row = grp * TR::RG + rqi * 4 + r;
            const int col = col0 + 4 * cs;
            if (qs == 0 && row < Nw && col < nc) {
                if (EPI == 1) {
                    p.w[0].out[(int64_t)(c0 + col) * p.w[0].ldo + row] = ps_silu_mul(yg[cs][r], v);
                } else {
                    float val = v;
                    if (b) val = __fadd_rn(val, b[row]);
                    if (p.residual && wi == 0) val = __fadd_rn(p.residual[(int64_t)(c0 + col) * ldo + row], val);
                    o[(int64_t)(c0 + col) * ldo + row] = val;
                }
            }
        }
    }
}

} // namespace

int psk_gemv_debug(int key, uint64_t *host_out, int n_words) {
    if (!host_out) {
        if (key >= 0 && !g_dbg_buf && hipMalloc((void **)&g_dbg_buf, (size_t)2 * G3_DBG_WGS * 64 * 8) != hipSuccess) return 2;
        if (key >= 0 && hipMemset(g_dbg_buf, 0, (size_t)2 * G3_DBG_WGS * 64 * 8) != hipSuccess) return 2;
        g_dbg_key = key;
        return 0;
    }
    if (!g_dbg_buf || n_words > 2 * G3_DBG_WGS * 64) return 1;
    return hipMemcpy(host_out, g_dbg_buf, (size_t)n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}

unsigned long long *psk_gemv_dbg_buf(int epi, int pro) { // key = k1 + 100 * (k2 + 1): launches matching k1 record into the first half, k2 into the second
    if (!g_dbg_buf || g_dbg_key < 0) return nullptr;
    const int k1 = g_dbg_key % 100, k2 = g_dbg_key / 100 - 1;
    if (k1 == epi * 4 + pro) return g_dbg_buf;
    if (k2 == epi * 4 + pro) return g_dbg_buf + (size_t)G3_DBG_WGS * 64;
    return nullptr;
}

bool psk_gemv_rope_ok(int wt, int64_t K) { // a Q / K / V launch of ONE type whose epilogue rotates and appends: mirrors psk_gemv4 / psk_gemvb / psk_gemvk
    if (wt == PS_Q4_K) return psk_gemv4_covers(K);
    if (psk_gemvb_covers(wt, K)) return true;
    return wt == PS_Q5_K && psk_gemvk_covers(wt, K); // (Q / K / V all Q5_K: k_gemvk.hip)
}

size_t psk_gemv_lds_col_bytes(int wt, int64_t K) {
    const int64_t blk = (wt == PS_Q4_K) ? 256 : 32, unit = (wt == PS_Q4_K) ? 256 : 128;
    const int64_t Kp = (K + unit - 1) / unit * unit;
    const size_t b = (size_t)Kp + (size_t)(Kp / blk) * 4 + (size_t)(Kp / 32) * 4 + (size_t)(Kp / 16) * 2;
    return (b + 15) / 16 * 16;
}

// Batched mat-mul from pre-quantized activations; returns -1 when the shape is not covered (caller falls back to
// column groups through the mat-vec).
template <int EPI, int NWV, int C>
static void launch_gemm8m_k(hipStream_t st, const GemvParams &p, const dim3 grid, size_t smem) {
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) { (void)hipFuncSetAttribute((const void *)gemm8m_kernel<EPI, NWV, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); }
    psk_note_kernel("gemm8m_kernel<%d, %d, %d>", EPI, NWV, C);
    hipLaunchKernelGGL((gemm8m_kernel<EPI, NWV, C>), grid, dim3(NWV * 64), smem, st, p);
}
template <int WT, int EPI, int NWV>
static void launch_gemm8b_k(hipStream_t st, const GemvParams &p, const dim3 grid, size_t smem) {
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) { (void)hipFuncSetAttribute((const void *)gemm8b_kernel<WT, EPI, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); }
    psk_note_kernel("gemm8b_kernel<%d, %d, %d>", WT, EPI, NWV);
    hipLaunchKernelGGL((gemm8b_kernel<WT, EPI, NWV>), grid, dim3(NWV * 64), smem, st, p);
}
template <int WT>
static int launch_gemm8(hipStream_t st, const GemvParams &p, int epi, int nwv, const dim3 grid, size_t smem) {
    if constexpr (WT != PS_Q4_K) { // quad dots on the matrix cores
        if (nwv == 8) launch_gemm8b_k<WT, 1, 8>(st, p, grid, smem); // (Q4_0 gate/up: see psk_gemm8)
        else if (nwv == 16) { if (epi) launch_gemm8b_k<WT, 1, 16>(st, p, grid, smem); else launch_gemm8b_k<WT, 0, 16>(st, p, grid, smem); }
        else { if (epi) launch_gemm8b_k<WT, 1, 4>(st, p, grid, smem); else launch_gemm8b_k<WT, 0, 4>(st, p, grid, smem); }
    } else {
        if (nwv == 8) { if (epi) launch_gemm8m_k<1, 8, 16>(st, p, grid, smem); else launch_gemm8m_k<0, 8, 16>(st, p, grid, smem); } // 16 columns
        else if (nwv == 16) { if (epi) launch_gemm8m_k<1, 16, 8>(st, p, grid, smem); else launch_gemm8m_k<0, 16, 8>(st, p, grid, smem); }
        else { if (epi) launch_gemm8m_k<1, 4, 8>(st, p, grid, smem); else launch_gemm8m_k<0, 4, 8>(st, p, grid, smem); }
    }
    return 0;
}
int psk_gemm8(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K, int64_t bs) {
    if (a.pro != 0 || a.n_w < 1) return -1;
    if (a.rope && !(a.w[0]->dtype == PS_Q4_K && bs >= 2)) return -1;
    if (a.w[0]->dtype == PS_Q4_K) { // prefill chunks: fp16 matrix cores on exact integers, producer / consumer waves (k_gemm4k.hip)
        const int rc = psk_gemm4k(st, n_cu, a, act, K, bs);
        if (rc != -1) return rc;
    }
    if (a.rope) return 3; // (a batch with the RoPE epilogue requested that the chunk mat-mul did not take: psk_gemm4k_rope_ok and psk_gemm4k disagree)
    const int wt = a.w[0]->dtype;
    if (wt != PS_Q4_K && wt != PS_Q8_0 && wt != PS_Q4_0) return -1;
    const int64_t unit = wt == PS_Q4_K ? 256 : 128, blk = wt == PS_Q4_K ? 256 : 32, rg = wt == PS_Q4_0 ? 16 : 8;
    if (K % unit) return -1;
    GemvParams p{};
    p.n_w = a.n_w; p.K = K; p.bs = bs; p.residual = a.residual;
    p.aq = act.qs; p.ad = act.d; p.abs16 = act.bs16;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != wt || a.w[i]->K != K) return -1;
        const int64_t ng = (a.w[i]->N + rg - 1) / rg;
        p.w[i] = GemvW{a.w[i]->qs, a.w[i]->aux, a.out[i], a.bias[i], a.w[i]->N, a.ldo[i], ng};
        p.groups_total += ng;
    }
    const int epi = a.silu_pair ? 1 : 0;
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N)) return -1;
    const size_t smem = (size_t)(K / unit) * 8 * (wt == PS_Q4_K ? 304 : 144); // [unit][column] records (gemm8_stage)
    if (smem > 158 * 1024) return -1;
    const int64_t n_tasks = epi == 1 ? p.w[0].n_groups : p.groups_total;
    // one wave per row group; 16-wave workgroups amortise the LDS staging of the 8 columns, 4-wave workgroups spread a
    // small launch (tree verify, short prefill tails) over all CUs
    int64_t ncg = (bs + 7) / 8;
    int nwv = ((n_tasks + 15) / 16) * ncg < (int64_t)n_cu ? 4 : 16;
    size_t smem_l = smem;
    // Q4_K, short rows, wide batches: 16 columns per workgroup of 8 waves (the weight-side work of a unit is shared by
    // twice the columns; the image of 16 columns has to fit the LDS)
    static const bool no16 = getenv("PS_GEMM8_C8") != nullptr;
    if (wt == PS_Q4_K && !no16 && nwv == 16 && bs > 8 && 2 * smem <= 80 * 1024) { nwv = 8; ncg = (bs + 15) / 16; smem_l = 2 * smem; }
    if (wt == PS_Q4_0 && epi == 1 && nwv == 16) nwv = 8; // two column sets + the held gate rows need more than the 128 VGPRs of a 16-wave workgroup
    const dim3 grid((unsigned)((n_tasks + nwv - 1) / nwv), (unsigned)ncg);
    switch (wt) {
    case PS_Q4_K: return launch_gemm8<PS_Q4_K>(st, p, epi, nwv, grid, smem_l);
    case PS_Q8_0: return launch_gemm8<PS_Q8_0>(st, p, epi, nwv, grid, smem_l);
    default: return launch_gemm8<PS_Q4_0>(st, p, epi, nwv, grid, smem_l);
    }
}

// up to psk_gemv_max_cols columns per launch (more than 4 only from pre-quantized activations); larger batches are
// split by the caller.
int psk_gemv_max_cols(int wt, int64_t K) {
    const size_t cb = psk_gemv_lds_col_bytes(wt, K);
    (void)cb;
    return 4; // (the 8-column instantiations of the in-lane kernel measured slower than 4; batches take psk_gemm8)
}

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
    if ((size_t)p.col_bytes * (bs == 1 ? 1 : 4) > 158 * 1024) return 7;
    const int epi = a.silu_pair ? 1 : (a.rope ? 2 : 0);
    const bool rope_part = a.rope && (a.n_w != 3 || a.rope_wi0 != 0); // a part of the Q / K / V triple: gemv4 only
    if (a.rope) {
        if (a.n_w + a.rope_wi0 > 3 || bs != 1 || a.pro != 1) return 9;
        p.rope = *a.rope;
    }
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N)) return 5;
    if (bs == 1 && wt == PS_Q4_K) { // second-generation decode kernel (k_gemv4.hip)
        const int rc = psk_gemv4(st, n_cu, a, act, K);
        if (rc != -1) return rc;
    }
    if (rope_part) return 9;
    if (bs == 1 && (wt == PS_Q4_0 || wt == PS_Q8_0)) { // producer / chain-wave kernel of the 32-element block formats (k_gemvb.hip)
        const int rc = psk_gemvb(st, n_cu, a, act, K);
        if (rc != -1) return rc;
    }
    switch (wt) {
    case PS_Q4_0: return launch_wt<PS_Q4_0>(st, n_cu, p, epi, a.pro);
    case PS_Q8_0: return launch_wt<PS_Q8_0>(st, n_cu, p, epi, a.pro);
    case PS_Q4_K: return launch_wt<PS_Q4_K>(st, n_cu, p, epi, a.pro);
    }
    return 6;
}
