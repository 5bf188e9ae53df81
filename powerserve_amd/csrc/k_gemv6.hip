// Q6_K and Q5_K weights x Q8_K activations: y[c][row] = ggml_vec_dot_q{6,5}_K_q8_K(K, W[row], act[c])  
//
// Reference numerics (AVX2 branch, libs/ggml/src/ggml-quants.c:9040-9115):
// per super-block of 256 weights an EXACT int32 vector sumi[8] is formed (lane u owns weights 4u..4u+3 of each of the
// eight 32-element sub-vectors, each 16-element half scaled by its int8 scale), then ONE fma per super-block and lane,
//   acc[u] = fma(d_w * d_act, (float)sumi[u], acc[u]),
// and hsum_float_8 over the lanes at the end.  So the fp32 chain is only K/256 steps long and everything else is
// integer work that may be done in any order.
//
// Mapping: one wave per weight row, lane = (sbl, u): eight consecutive super-blocks x the eight AVX lanes.  The device
// layout (ps_internal.h) makes the wave's ql / qh fetch one contiguous 1 KiB / 512 B request per 8 super-blocks.  Every
// lane computes its sumi with v_dot4; the chain is then walked in super-block order with two lane broadcasts per step
// (all 64 lanes redundantly carry acc[u] of their own u, so no second exchange is needed before the lane reduction).
// HBM-bound: 210 B per 256 weights, no reuse; activations (K bytes per column) stay in L1/L2.
#include "ps_dev.h"
#include "ps_internal.h"

namespace {

struct Gemv6Params {
    const uint8_t *ql, *qh;
    const uint8_t *sc;
    const uint16_t *d;
    int64_t K, N;
    int nsb;
    const int8_t *aq;  // [bs][K]
    const float *ad;   // [bs][K/256]
    float *out;        // [bs][ldo]
    int64_t ldo;
    const float *bias;     // [N] or null
    const float *residual; // [bs][ldo] or null
    int nc;                // live columns (<= BS)
};

__device__ __forceinline__ int sbyte(uint32_t v, int byte) { return (int)(v << (24 - 8 * byte)) >> 24; }

template <int BS>
__global__ __launch_bounds__(256) void gemv6_kernel(Gemv6Params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sbl = lane >> 3, u = lane & 7, hi = u >> 2;
    const int nsb = p.nsb, nit = (nsb + 7) >> 3;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.N; row += (int64_t)gridDim.x * 4) {
        float acc[BS];
#pragma unroll
        for (int c = 0; c < BS; c++) acc[c] = 0.f;
        const int64_t rb = row * nsb;
        for (int it = 0; it < nit; it++) {
            const int sb0 = it * 8, sbr = sb0 + sbl;
            const int sb = sbr < nsb ? sbr : nsb - 1; // lanes past the row end re-read the last block; never chained
            const uint4 L = ld_stream16(p.ql + (rb + sb) * 128 + u * 16);
            const uint2 H = *(const uint2 *)(p.qh + (rb + sb) * 64 + u * 8);
            const uint4 S = *(const uint4 *)(p.sc + (rb + sb) * 16);
            const float dw = ps_h2f(p.d[rb + sb]);
            // 6-bit weights of this lane: q[j][sub] holds weights 4u..4u+3 of sub-vector 4j+sub as bytes 0..63
            uint32_t q[2][4];
            {
                const uint32_t A0 = L.x, B0 = L.y, A1 = L.z, B1 = L.w;
                q[0][0] = (A0 & 0x0F0F0F0Fu) | ((H.x & 0x03030303u) << 4);
                q[0][1] = (B0 & 0x0F0F0F0Fu) | (((H.x >> 2) & 0x03030303u) << 4);
                q[0][2] = ((A0 >> 4) & 0x0F0F0F0Fu) | (((H.x >> 4) & 0x03030303u) << 4);
                q[0][3] = ((B0 >> 4) & 0x0F0F0F0Fu) | (((H.x >> 6) & 0x03030303u) << 4);
                q[1][0] = (A1 & 0x0F0F0F0Fu) | ((H.y & 0x03030303u) << 4);
                q[1][1] = (B1 & 0x0F0F0F0Fu) | (((H.y >> 2) & 0x03030303u) << 4);
                q[1][2] = ((A1 >> 4) & 0x0F0F0F0Fu) | (((H.y >> 4) & 0x03030303u) << 4);
                q[1][3] = ((B1 >> 4) & 0x0F0F0F0Fu) | (((H.y >> 6) & 0x03030303u) << 4);
            }
            // scale of (j, sub) for this lane: scales[8j + 2 sub + (u >= 4)]
            int scl[2][4];
            {
                const uint32_t Sw[4] = {S.x, S.y, S.z, S.w};
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    scl[j][0] = sbyte(Sw[2 * j], hi);
                    scl[j][1] = sbyte(Sw[2 * j], 2 + hi);
                    scl[j][2] = sbyte(Sw[2 * j + 1], hi);
                    scl[j][3] = sbyte(Sw[2 * j + 1], 2 + hi);
                }
            }
            const int nlive = nsb - sb0 < 8 ? nsb - sb0 : 8; // wave-uniform
#pragma unroll
            for (int c = 0; c < BS; c++) {
                const int cc = c < p.nc ? c : p.nc - 1;
                const int8_t *a = p.aq + (int64_t)cc * p.K + (int64_t)sb * 256 + 4 * u;
                int sumi = 0;
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int sub = 0; sub < 4; sub++) {
                        const int av = *(const int *)(a + j * 128 + sub * 32);
                        // sum (q - 32) * a : maddubs(q, a) - maddubs(32, a) of the reference, exact in int32
                        const int s = dot4((int)q[j][sub], av, 0) - dot4(0x20202020, av, 0);
                        sumi += scl[j][sub] * s;
                    }
                const float t  = (float)sumi;
                const float dd = __fmul_rn(p.ad[(int64_t)cc * nsb + sb], dw); // y[i].d * GGML_FP16_TO_FP32(x[i].d)
                float ds[8], ts[8]; // all sixteen lane broadcasts in flight, then the chain in super-block order
#pragma unroll
                for (int s = 0; s < 8; s++) { ds[s] = __shfl(dd, s * 8 + u, 64); ts[s] = __shfl(t, s * 8 + u, 64); }
                float ac = acc[c];
#pragma unroll
                for (int s = 0; s < 8; s++)
                    if (s < nlive) ac = __fmaf_rn(ds[s], ts[s], ac);
                acc[c] = ac;
            }
        }
#pragma unroll
        for (int c = 0; c < BS; c++) { // hsum_float_8 (ggml-quants.c:62-68): (a4+a0, a5+a1, a6+a2, a7+a3) -> (r0+r2, r1+r3) -> sum
            float v = acc[c];
            v = __fadd_rn(v, dpp_f<0x104>(v)); // one-directional row shifts: lane 0 of each 8-group ends with the reference's sum
            v = __fadd_rn(v, dpp_f<0x102>(v));
            v = __fadd_rn(v, dpp_f<0x101>(v));
            if (lane == 0 && c < p.nc) {
                if (p.bias) v = __fadd_rn(v, p.bias[row]);
                if (p.residual) v = __fadd_rn(p.residual[(int64_t)c * p.ldo + row], v);
                p.out[(int64_t)c * p.ldo + row] = v;
            }
        }
    }
}

// ---------------------------------------------------------------- Q5_K (ggml-quants.c:8382-8459, AVX2 branch)
// Same wave mapping and the same integer/fp32 split as Q6_K: lane (sbl, u) forms the exact int32
//   sumi[u] = sum over the 8 sub-vectors j of  scale[j] * dot4(q5[j][4u..4u+3], y[j][4u..4u+3]),   q5 = low nibble + 16 * (bit j of qh),
// and acc[u] = fma(d_w * d_act, (float)sumi[u], acc[u]) is walked in super-block order.  The mins do not ride in vector
// lanes in this kernel of the reference but in ONE scalar,  summs += dmin * (float)hsum(mins . bsums)  — a multiply and an
// add (two roundings: the reference build this backend is pinned to has fp-contraction off) — so a
// third value per super-block joins the two that are broadcast for the chain.  Result: hsum_float_8(acc) + summs.
struct Gemv5Mat { // one weight matrix of a launch (up to three share the activation: Q / K / V, gate / up)
    const uint8_t *qs, *qh;
    const uint4 *hdr;     // [N][K/256] {d | dmin << 16, scales[12]}
    int64_t N, ldo;
    float *out;
    const float *bias, *residual;
};
struct Gemv5Params {
    Gemv5Mat w[3];
    int n_w;
    int64_t K, N;         // N: rows of all matrices together
    int nsb;
    const int8_t *aq;     // [bs][K]
    const float *ad;      // [bs][K/256]
    const int16_t *abs16; // [bs][K/16]
    int nc;
};

template <int BS>
__global__ __launch_bounds__(256) void gemv5_kernel(Gemv5Params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sbl = lane >> 3, u = lane & 7;
    const int nsb = p.nsb, nit = (nsb + 7) >> 3;
    for (int64_t grow = (int64_t)blockIdx.x * 4 + wave; grow < p.N; grow += (int64_t)gridDim.x * 4) {
        int64_t row = grow; // the wave's matrix and its row in it
        int wi = 0;
        if (p.n_w > 1 && row >= p.w[0].N) { row -= p.w[0].N; wi = 1; }
        if (p.n_w > 2 && wi == 1 && row >= p.w[1].N) { row -= p.w[1].N; wi = 2; }
        const Gemv5Mat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
        float acc[BS], summs[BS];
#pragma unroll
        for (int c = 0; c < BS; c++) { acc[c] = 0.f; summs[c] = 0.f; }
        const int64_t rb = row * nsb;
        for (int it = 0; it < nit; it++) {
            const int sb0 = it * 8, sbr = sb0 + sbl;
            const int sb = sbr < nsb ? sbr : nsb - 1; // lanes past the row end re-read the last block; never chained
            const uint4 Q = ld_stream16(W.qs + (rb + sb) * 128 + u * 16);
            const uint32_t H = *(const uint32_t *)(W.qh + (rb + sb) * 32 + u * 4);
            const uint4 hd = W.hdr[rb + sb];
            const float dw = ps_h2f((uint16_t)(hd.x & 0xffff)), dmw = ps_h2f((uint16_t)(hd.x >> 16));
            // 5-bit weights of this lane: q[j] = elements 4u..4u+3 of sub-vector j, one per byte
            const uint32_t Qw[4] = {Q.x, Q.y, Q.z, Q.w};
            uint32_t q[8];
            int scl[8], mnl[8];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                q[2 * jj]     = (Qw[jj] & 0x0F0F0F0Fu) | (((H >> (2 * jj)) & 0x01010101u) << 4);
                q[2 * jj + 1] = ((Qw[jj] >> 4) & 0x0F0F0F0Fu) | (((H >> (2 * jj + 1)) & 0x01010101u) << 4);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) ps_scale_min_k4(j, hd.y, hd.z, hd.w, scl[j], mnl[j]);
            const int nlive = nsb - sb0 < 8 ? nsb - sb0 : 8; // wave-uniform
#pragma unroll
            for (int c = 0; c < BS; c++) {
                const int cc = c < p.nc ? c : p.nc - 1;
                const int8_t *a = p.aq + (int64_t)cc * p.K + (int64_t)sb * 256 + 4 * u;
                int sumi = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) sumi += scl[j] * dot4((int)q[j], *(const int *)(a + j * 32), 0);
                // mins . (bsums[2j] + bsums[2j+1]): every lane of the super-block forms the same int32
                const int16_t *bs = p.abs16 + ((int64_t)cc * nsb + sb) * 16;
                int hsum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) hsum += mnl[j] * (int)(int16_t)(bs[2 * j] + bs[2 * j + 1]);
                const float yd = p.ad[(int64_t)cc * nsb + sb];
                const float dd = __fmul_rn(yd, dw);                                  // y[i].d * GGML_FP16_TO_FP32(x[i].d)
                const float mm = __fmul_rn(__fmul_rn(-yd, dmw), (float)hsum);        // dmin * hsum, dmin = -y[i].d * fp16(x[i].dmin)
                const float t  = (float)sumi;
                float ds[8], ts[8], ms[8]; // all lane broadcasts in flight, then the chains in super-block order
#pragma unroll
                for (int s = 0; s < 8; s++) { ds[s] = __shfl(dd, s * 8 + u, 64); ts[s] = __shfl(t, s * 8 + u, 64); ms[s] = __shfl(mm, s * 8 + u, 64); }
                float ac = acc[c], sm = summs[c];
#pragma unroll
                for (int s = 0; s < 8; s++)
                    if (s < nlive) { ac = __fmaf_rn(ds[s], ts[s], ac); sm = __fadd_rn(sm, ms[s]); }
                acc[c] = ac; summs[c] = sm;
            }
        }
#pragma unroll
        for (int c = 0; c < BS; c++) { // hsum_float_8 (ggml-quants.c:62-68), then + summs
            float v = acc[c];
            v = __fadd_rn(v, dpp_f<0x104>(v));
            v = __fadd_rn(v, dpp_f<0x102>(v));
            v = __fadd_rn(v, dpp_f<0x101>(v));
            v = __fadd_rn(v, summs[c]);
            if (lane == 0 && c < p.nc) {
                if (W.bias) v = __fadd_rn(v, W.bias[row]);
                if (W.residual) v = __fadd_rn(W.residual[(int64_t)c * W.ldo + row], v);
                W.out[(int64_t)c * W.ldo + row] = v;
            }
        }
    }
}

int launch_gemv5(hipStream_t st, int n_cu, const psk_gemv6_args *a, int n_w, ps_act act, int64_t K, int64_t bs) {
    Gemv5Params p{};
    p.n_w = n_w; p.K = K; p.nsb = (int)(K / 256);
    for (int i = 0; i < n_w; i++) p.N += a[i].w->N;
    const int64_t nwg = (p.N + 3) / 4;
    const unsigned grid = (unsigned)(nwg < (int64_t)n_cu * 16 ? nwg : (int64_t)n_cu * 16);
    for (int64_t c0 = 0; c0 < bs; c0 += 8) {
        const int nc = (int)(bs - c0 < 8 ? bs - c0 : 8);
        p.aq = act.qs + c0 * K; p.ad = act.d + c0 * (K / 256); p.abs16 = act.bs16 + c0 * (K / 16);
        for (int i = 0; i < n_w; i++) {
            const ps_weight *w = a[i].w;
            p.w[i] = Gemv5Mat{w->qs, w->qh, (const uint4 *)w->sc, w->N, a[i].ldo, a[i].out + c0 * a[i].ldo, a[i].bias,
                              a[i].residual ? a[i].residual + c0 * a[i].ldo : nullptr};
        }
        p.nc = nc;
        if (nc == 1) { psk_note_kernel("gemv5_kernel<1>"); hipLaunchKernelGGL(gemv5_kernel<1>, dim3(grid), dim3(256), 0, st, p); }
        else if (nc == 2) { psk_note_kernel("gemv5_kernel<2>"); hipLaunchKernelGGL(gemv5_kernel<2>, dim3(grid), dim3(256), 0, st, p); }
        else if (nc <= 4) { psk_note_kernel("gemv5_kernel<4>"); hipLaunchKernelGGL(gemv5_kernel<4>, dim3(grid), dim3(256), 0, st, p); }
        else { psk_note_kernel("gemv5_kernel<8>"); hipLaunchKernelGGL(gemv5_kernel<8>, dim3(grid), dim3(256), 0, st, p); }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

} // namespace

int psk_gemv6(hipStream_t st, int n_cu, const psk_gemv6_args &a, ps_act act, int64_t K, int64_t bs) {
    const ps_weight *w = a.w;
    if (w->dtype == PS_Q5_K && w->K == K && K % 256 == 0) {
        if (bs >= ps_gemm4k_min_cols()) { // batches: the matrix-core mat-mul with the Q5_K producer (k_gemm4k.hip)
            const int rc = psk_gemm5k(st, n_cu, a, act, K, bs);
            if (rc != -1) return rc;
        }
        return launch_gemv5(st, n_cu, &a, 1, act, K, bs);
    }
    if (w->dtype != PS_Q6_K || w->K != K || K % 256) return 4;
    { // chunks and wide trees: the matrix-core mat-mul (k_gemm4k.hip)
        const int rc = psk_gemm6k(st, n_cu, a, act, K, bs);
        if (rc != -1) return rc;
    }
    Gemv6Params p{};
    p.ql = w->qs; p.qh = w->qh; p.sc = w->sc; p.d = (const uint16_t *)w->aux;
    p.K = K; p.N = w->N; p.nsb = (int)(K / 256);
    p.ldo = a.ldo; p.bias = a.bias;
    const int64_t nwg = (w->N + 3) / 4;
    const unsigned grid = (unsigned)(nwg < (int64_t)n_cu * 16 ? nwg : (int64_t)n_cu * 16);
    for (int64_t c0 = 0; c0 < bs; c0 += 8) {
        const int nc = (int)(bs - c0 < 8 ? bs - c0 : 8);
        p.aq = act.qs + c0 * K; p.ad = act.d + c0 * (K / 256);
        p.out = a.out + c0 * a.ldo; p.residual = a.residual ? a.residual + c0 * a.ldo : nullptr; p.nc = nc;
        if (nc == 1) { psk_note_kernel("gemv6_kernel<1>"); hipLaunchKernelGGL(gemv6_kernel<1>, dim3(grid), dim3(256), 0, st, p); }
        else if (nc == 2) { psk_note_kernel("gemv6_kernel<2>"); hipLaunchKernelGGL(gemv6_kernel<2>, dim3(grid), dim3(256), 0, st, p); }
        else if (nc <= 4) { psk_note_kernel("gemv6_kernel<4>"); hipLaunchKernelGGL(gemv6_kernel<4>, dim3(grid), dim3(256), 0, st, p); }
        else { psk_note_kernel("gemv6_kernel<8>"); hipLaunchKernelGGL(gemv6_kernel<8>, dim3(grid), dim3(256), 0, st, p); }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Q5_K matrices that share one activation (Q / K / V, gate / up) in ONE launch: single tokens and batches the chunk kernels do
// not take.  4: not all Q5_K of this K.
int psk_gemv5_multi(hipStream_t st, int n_cu, const psk_gemv6_args *a, int n_w, ps_act act, int64_t K, int64_t bs) {
    if (n_w < 1 || n_w > 3 || K % 256) return 4;
    for (int i = 0; i < n_w; i++)
        if (!a[i].w || a[i].w->dtype != PS_Q5_K || a[i].w->K != K) return 4;
    return launch_gemv5(st, n_cu, a, n_w, act, K, bs);
}
