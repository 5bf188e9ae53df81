// Decode mat-vec for the K-quants that are stored as per-row planes (Q6_K, Q5_K weights x one Q8_K activation column): the
// producer / chain-wave form of gemv4 (k_gemv4.hip) for ggml_vec_dot_q6_K_q8_K (libs/ggml/src/ggml-quants.c:9040-9115) and
// ggml_vec_dot_q5_K_q8_K (:8382-8459), bit-exact (the numerics contract is k_gemv6.hip's: per super-block an exact int32
// sumi[u], ONE fma  acc[u] = fma(d_w * d_y, (float)sumi[u], acc[u])  per super-block and AVX lane, hsum_float_8 at the end; Q5_K
// adds the scalar  summs += dmin * (float)(mins . bsums)  as a multiply and an add).
//
// Until round 3 these types ran one wave per weight row with the activation quantized by its own launch and nothing fused
// (k_gemv6.hip: 1.4-1.6 TB/s; Q5_K_M decoded at 293 tok/s against 513 for pure Q4_K).  Here:
//   * a unit = one super-block of a group of 8 rows, lane = (row r, AVX lane u) as everywhere; the lane's bytes come straight
//     from the row planes (8 full 128-byte lines per wave and 16-byte load), no repack;
//   * eight producer waves turn units into records {d_w * d_y, (float)sumi} (+ dmin * hsum for Q5_K), ONE chain wave does the
//     fmas in super-block order; the activation row is requested first, RMSNorm + Q8_K quantization run once per workgroup in
//     the prologue (quad-major tiles in LDS: a lane's eight activation dwords of a unit are two ds_read_b128);
//   * Q6_K's  sum (q - 32) y  needs no correction term: (q ^ 32) << 2 IS the signed byte 4 (q - 32), v_dot4 returns four
//     times the sum and the unit's int32 is shifted back once;
//   * a register ring of DC chunks of UPW units per producer (default 3 chunks of 2: 80 KB of Q6_K in flight per CU), the first chunk
//     requested before the prologue, the others behind the quantizer.
// Epilogues as gemv4: EPI 0 bias / residual, EPI 1 SiLU(gate) * up, EPI 2 adjacent-pair RoPE + KV-cache append (Q / K / V).
#include "ps_gemv_dev.h"

namespace {

template <int WT> struct GKUnit;
template <> struct GKUnit<PS_Q6_K> { ps_u32x4 L, S; uint32_t H0, H1, d; };
template <> struct GKUnit<PS_Q5_K> { ps_u32x4 Q; uint32_t H; ps_u32x4 hd; };
template <int WT> struct GKRec { using T = float2; };
template <> struct GKRec<PS_Q5_K> { using T = float4; };

struct GKMat {
    const uint8_t *qs, *qh, *sc; // Q6_K: ql / qh / int8 scales;  Q5_K: qs / qh / headers {d | dmin << 16, scales[12]}
    const uint16_t *d;           // Q6_K: fp16 d per (row, super-block)
    float *out;
    const float *bias;
    int64_t N;
    int n_groups;
};
struct GKParams {
    GKMat w[3];
    int n_w, n_units, n_tasks;  // tasks: row groups (EPI 0 / 2) or gate/up row-group pairs (EPI 1)
    int split_q, split_r;       // tasks per workgroup = split_q (+1 for the first split_r workgroups)
    int K, act_bytes;
    const float *residual;
    const float *x, *nw;        // PRO 1: rmsnorm(x, nw, eps) then quantize;  PRO 2: quantize(x)
    float eps;
    const int8_t *aq;           // PRO 0: activation already quantized
    const float *ad;
    const int16_t *abs16;
    psk_rope_kv rope;           // EPI 2
    int rope_wi0;               // EPI 2: w[0]'s place in the Q / K / V triple (a launch may carry a part of it)
};

// NW producer waves, UPW units per producer and chunk, DC chunks in flight per producer (EA of them requested before the
// prologue), TPW activation tiles per producer
template <int WT, int NW, int UPW, int DC, int EA, int TPW, int EPI, int PRO>
__global__ __launch_bounds__((NW + 1) * 64) void gemvk_kernel(const GKParams p) {
    using Rec = typename GKRec<WT>::T;
    constexpr int UPB = NW * UPW;
    constexpr bool MULTI = WT != PS_Q6_K; // several matrices per launch (Q6_K launches carry one: no per-lane matrix selects)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ uint64_t exp_tab[PS_EXP2F_N];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K, n_units = p.n_units;
    int8_t *lq = (int8_t *)smem;              // [K] quants, quad-major inside a 256-element tile (dword u * 8 + g)
    float *ld  = (float *)(smem + K);         // [K / 256] tile scales
    int *lb    = (int *)(ld + n_units);       // [K / 32] sums of 32 (Q5_K's mins term)
    Rec *recs  = (Rec *)(smem + p.act_bytes); // [2][UPB][64]
    float *epA = (float *)(recs + 2 * UPB * 64);
    const int r = lane >> 3, u = lane & 7;

    const int tot = (EPI == 1) ? 2 * n_units : n_units; // stream units per task (EPI 1: gate units then up units)
    const int t0  = (int)blockIdx.x * p.split_q + min((int)blockIdx.x, p.split_r);
    const int nt  = p.split_q + ((int)blockIdx.x < p.split_r ? 1 : 0);
    const int s_end    = nt * tot;
    const int n_chunks = (s_end + UPB - 1) / UPB;
    const int n_iters  = (n_chunks + DC - 1) / DC;

    if (wave < NW) { // ------------------------------------------------------------------ producers
        // (two tiles per wave-instruction, g4_quantize_pair: pair tp = wave + i * NW, lane l holds elements tp * 512 + 8 l .. + 7; an odd row's last
        //  pair has one tile: its upper lanes load element 0 and store nothing)
        constexpr int PPW = (TPW + 1) / 2;
        const int n_pairs = (n_units + 1) / 2;
        float4 xv[PPW][2], wv[PPW][2];
        if (PRO != 0) {
#pragma unroll
            for (int i = 0; i < PPW; i++) {
                const int tp = wave + i * NW;
                int64_t e = (int64_t)(tp < n_pairs ? tp : 0) * 512 + lane * 8;
                e = e < K ? e : 0; // (never a branch around a load)
                xv[i][0] = *(const float4 *)(p.x + e); xv[i][1] = *(const float4 *)(p.x + e + 4);
                if (PRO == 1) { wv[i][0] = *(const float4 *)(p.nw + e); wv[i][1] = *(const float4 *)(p.nw + e + 4); }
            }
        }
        int step_t = 0, step_u = DC * UPB; // a trip moves a unit of the ring DC chunks on
        while (step_u >= tot) { step_u -= tot; step_t++; }
        int tS[DC][UPW], uS[DC][UPW];
#pragma unroll
        for (int d = 0; d < DC; d++)
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                int t = 0, un = d * UPB + wave * UPW + i;
                while (un >= tot) { un -= tot; t++; }
                tS[d][i] = t; uS[d][i] = un;
            }
        GKUnit<WT> ring[DC][UPW];
        // loads are UNCONDITIONAL (a unit past the range re-reads the workgroup's first unit) so that the compiler counts vmcnt
        // exactly and a chunk is consumed while the next ones are in flight
        auto issue = [&](const int d) {
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                const bool live = tS[d][i] < nt;
                int grp = t0 + (live ? tS[d][i] : 0), ul = live ? uS[d][i] : 0, wi = 0;
                if (EPI == 1) {
                    if (ul >= n_units) { ul -= n_units; wi = 1; }
                } else if (MULTI && p.n_w > 1 && grp >= p.w[0].n_groups) {
                    grp -= p.w[0].n_groups; wi = 1;
                    if (p.n_w > 2 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
                }
                const uint8_t *qs = p.w[0].qs, *qh = p.w[0].qh, *sc = p.w[0].sc;
                const uint16_t *dp = p.w[0].d;
                if (wi == 1) { qs = p.w[1].qs; qh = p.w[1].qh; sc = p.w[1].sc; dp = p.w[1].d; }
                if (wi == 2) { qs = p.w[2].qs; qh = p.w[2].qh; sc = p.w[2].sc; dp = p.w[2].d; }
                const uint64_t rsb = (uint64_t)(grp * 8 + r) * (uint64_t)n_units + (uint64_t)ul; // (row, super-block)
                GKUnit<WT> &U = ring[d][i];
                if constexpr (WT == PS_Q6_K) {
                    U.L = __builtin_nontemporal_load((const ps_u32x4 *)(qs + rsb * 128 + u * 16));
                    const ps_u32x2 hh = __builtin_nontemporal_load((const ps_u32x2 *)(qh + rsb * 64 + u * 8));
                    U.H0 = hh.x; U.H1 = hh.y;
                    U.S = *(const ps_u32x4 *)(sc + rsb * 16);
                    U.d = (uint32_t)dp[rsb];
                } else {
                    U.Q  = __builtin_nontemporal_load((const ps_u32x4 *)(qs + rsb * 128 + u * 16));
                    U.H  = __builtin_nontemporal_load((const uint32_t *)(qh + rsb * 32 + u * 4));
                    U.hd = *(const ps_u32x4 *)(sc + rsb * 16);
                }
            }
        };
        auto produce = [&](const int d, const int buf) {
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                if (tS[d][i] >= nt) continue; // wave-uniform: past this workgroup's range
                const int un = uS[d][i], ul = (EPI == 1 && un >= n_units) ? un - n_units : un;
                const GKUnit<WT> &U = ring[d][i];
                const int4 y0 = *(const int4 *)(lq + ul * 256 + u * 32), y1 = *(const int4 *)(lq + ul * 256 + u * 32 + 16); // sub-vectors 0..3, 4..7, quad u
                const float yd = ld[ul];
                Rec *out = recs + (size_t)(buf * UPB + wave * UPW + i) * 64 + lane;
                if constexpr (WT == PS_Q6_K) {
                    // ql bytes of the lane: L.x = bytes 4u.. of ql[0..31] (low nibble: sub-vector 0, high: sub-vector 2), L.y = of ql[32..63]
                    // (1, 3), L.z / L.w the same of the second half (4, 6 / 5, 7); qh: H.x, H.y = bytes 4u.. of the two halves, two bits per
                    // sub-vector.  Byte (q ^ 32) << 2 = 4 (q - 32) as int8.
                    constexpr uint32_t X = 0x80808080u;
                    const uint32_t h0 = U.H0, h1 = U.H1;
                    const int q0 = (int)((((U.L.x & 0x0F0F0F0Fu) << 2) | ((h0 & 0x03030303u) << 6)) ^ X);
                    const int q1 = (int)((((U.L.y & 0x0F0F0F0Fu) << 2) | ((h0 & 0x0C0C0C0Cu) << 4)) ^ X);
                    const int q2 = (int)((((U.L.x & 0xF0F0F0F0u) >> 2) | ((h0 & 0x30303030u) << 2)) ^ X);
                    const int q3 = (int)((((U.L.y & 0xF0F0F0F0u) >> 2) | (h0 & 0xC0C0C0C0u)) ^ X);
                    const int q4 = (int)((((U.L.z & 0x0F0F0F0Fu) << 2) | ((h1 & 0x03030303u) << 6)) ^ X);
                    const int q5 = (int)((((U.L.w & 0x0F0F0F0Fu) << 2) | ((h1 & 0x0C0C0C0Cu) << 4)) ^ X);
                    const int q6 = (int)((((U.L.z & 0xF0F0F0F0u) >> 2) | ((h1 & 0x30303030u) << 2)) ^ X);
                    const int q7 = (int)((((U.L.w & 0xF0F0F0F0u) >> 2) | (h1 & 0xC0C0C0C0u)) ^ X);
                    int sa[4], sb[4];
                    dot4x4(sa, q0, q1, q2, q3, y0.x, y0.y, y0.z, y0.w);
                    dot4x4(sb, q4, q5, q6, q7, y1.x, y1.y, y1.z, y1.w);
                    // scale of sub-vector g for this lane's half of it: scales[2 g + (u >= 4)]  (g = 4 j + sub: scales[8 j + 2 sub + hi])
                    const int sh = (u >> 2) * 8;
                    const uint32_t Sw[4] = {U.S.x, U.S.y, U.S.z, U.S.w};
                    int sumi = 0;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        sumi += __mul24(__builtin_amdgcn_sbfe((int)Sw[g >> 1], sh + 16 * (g & 1), 8), sa[g]);
                        sumi += __mul24(__builtin_amdgcn_sbfe((int)Sw[2 + (g >> 1)], sh + 16 * (g & 1), 8), sb[g]);
                    }
                    *out = make_float2(__fmul_rn(yd, ps_h2f((uint16_t)U.d)), (float)(sumi >> 2));
                } else {
                    constexpr uint32_t M = 0x0F0F0F0Fu;
                    const uint32_t Qw[4] = {U.Q.x, U.Q.y, U.Q.z, U.Q.w}, H = U.H;
                    int ql[4], qhh[4]; // sub-vectors 2 jj (low nibbles) and 2 jj + 1 (high), fifth bit = bit (sub-vector) of the qh byte
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        ql[jj]  = (int)((Qw[jj] & M) | (((H >> (2 * jj)) & 0x01010101u) << 4));
                        qhh[jj] = (int)(((Qw[jj] >> 4) & M) | (((H >> (2 * jj + 1)) & 0x01010101u) << 4));
                    }
                    int dlo[4], dhi[4];
                    dot4x4(dlo, ql[0], ql[1], ql[2], ql[3], y0.x, y0.z, y1.x, y1.z);   // sub-vectors 0, 2, 4, 6
                    dot4x4(dhi, qhh[0], qhh[1], qhh[2], qhh[3], y0.y, y0.w, y1.y, y1.w); // 1, 3, 5, 7
                    // get_scale_min_k4 (ggml-quants.c:1912-1920) on the 12 bytes hd.y, hd.z, hd.w
                    const uint32_t sc03 = U.hd.y & 0x3f3f3f3fu, sc47 = (U.hd.w & 0x0f0f0f0fu) | (((U.hd.y >> 6) & 0x03030303u) << 4);
                    const uint32_t mn03 = U.hd.z & 0x3f3f3f3fu, mn47 = ((U.hd.w >> 4) & 0x0f0f0f0fu) | (((U.hd.z >> 6) & 0x03030303u) << 4);
                    const uint32_t scv[4] = {__builtin_amdgcn_perm(0u, sc03, 0x0c010c00u), __builtin_amdgcn_perm(0u, sc03, 0x0c030c02u),
                                             __builtin_amdgcn_perm(0u, sc47, 0x0c010c00u), __builtin_amdgcn_perm(0u, sc47, 0x0c030c02u)};
                    int s = 0;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) // |dot4| <= 4 * 31 * 127 fits int16: {dl, dh} meet their scale pair in one v_dot2_i32_i16 (exact)
                        s = dot2_i16(__builtin_amdgcn_perm((uint32_t)dhi[jj], (uint32_t)dlo[jj], 0x05040100u), scv[jj], s);
                    // mins . (sums of 32): lane u contributes sub-vector u, the eight lanes of the row add up (integers: any order)
                    const uint32_t mw = u < 4 ? mn03 : mn47;
                    int hs = __mul24((int)((mw >> (8 * (u & 3))) & 0xffu), lb[ul * 8 + u]);
                    hs += dpp_i<0xB1>(hs); hs += dpp_i<0x4E>(hs); hs += dpp_i<0x141>(hs);
                    const float dw = ps_h2f((uint16_t)(U.hd.x & 0xffff)), dmw = ps_h2f((uint16_t)(U.hd.x >> 16));
                    *out = make_float4(__fmul_rn(yd, dw), (float)s, __fmul_rn(__fmul_rn(-yd, dmw), (float)hs), 0.f);
                }
                __builtin_amdgcn_sched_barrier(0); // one unit at a time
            }
        };
        auto advance = [&](const int d) {
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                tS[d][i] += step_t; uS[d][i] += step_u;
                if (uS[d][i] >= tot) { uS[d][i] -= tot; tS[d][i]++; }
            }
        };
        constexpr int EARLY = PRO == 0 ? DC : EA;
#pragma unroll
        for (int d = 0; d < EARLY; d++) issue(d);
        if (PRO == 0) {
            for (int i = threadIdx.x; i < K / 4; i += NW * 64) { // (quad-major tiles, as the quantizer writes them)
                const int dw = i & 63;
                ((int *)lq)[(i & ~63) + (((dw & 7) << 3) | (dw >> 3))] = ((const int *)p.aq)[i];
            }
            for (int i = threadIdx.x; i < n_units; i += NW * 64) ld[i] = p.ad[i];
            for (int i = threadIdx.x; i < K / 32; i += NW * 64) lb[i] = (int)p.abs16[2 * i] + (int)p.abs16[2 * i + 1];
        } else {
            // RMSNorm (PRO 1: ggml.c:12667-12720, double sum of squares, scale = 1/sqrtf(mean + eps), y = x * (w * scale)) and the
            // Q8_K quantization of this wave's tiles (tile t = wave + i * NW)
            float scale = 1.0f;
            if (PRO == 1) {
                double ss = 0.0;
#pragma unroll
                for (int i = 0; i < PPW; i++) {
                    const int tp = wave + i * NW;
                    if (tp < n_pairs && (int64_t)tp * 512 + lane * 8 < K) { // (lanes of a tile that does not exist add nothing)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            ss += (double)__fmul_rn(xv[i][h].x, xv[i][h].x);
                            ss += (double)__fmul_rn(xv[i][h].y, xv[i][h].y);
                            ss += (double)__fmul_rn(xv[i][h].z, xv[i][h].z);
                            ss += (double)__fmul_rn(xv[i][h].w, xv[i][h].w);
                        }
                    }
                }
                ss = wave_sum_d_dpp(ss);
                if (lane == 0) red[wave] = ss;
                __syncthreads();
                double tot_ss = 0.0;
#pragma unroll
                for (int i = 0; i <= NW; i++) tot_ss += red[i];
                const float mean = (float)(tot_ss / (double)K);
                scale            = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, p.eps)));
            }
#pragma unroll
            for (int i = 0; i < PPW; i++) {
                const int tp = wave + i * NW;
                const bool live = tp < n_pairs; // wave-uniform; a dead pair runs on pair 0's values and stores nothing
                float v[8] = {xv[i][0].x, xv[i][0].y, xv[i][0].z, xv[i][0].w, xv[i][1].x, xv[i][1].y, xv[i][1].z, xv[i][1].w};
                if (PRO == 1) {
                    const float w8[8] = {wv[i][0].x, wv[i][0].y, wv[i][0].z, wv[i][0].w, wv[i][1].x, wv[i][1].y, wv[i][1].z, wv[i][1].w};
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = __fmul_rn(v[k], __fmul_rn(w8[k], scale));
                }
                g4_quantize_pair(v, tp * 512 + lane * 8, tp, lq, ld, lb, live, n_units);
            }
#pragma unroll
            for (int d = EARLY; d < DC; d++) issue(d);
        }
        __syncthreads();
        for (int it = 0; it < n_iters; it++) { // chunk it * DC + d from ring slot d
#pragma unroll
            for (int d = 0; d < DC; d++) {
                produce(d, (it * DC + d) & 1);
                advance(d);
                issue(d);
                __syncthreads();
            }
        }
    } else { // ------------------------------------------------------------------------- chain wave
        const int ep_n = (p.split_q + 1) * 8;
        float *const epB = epA + ep_n, *const epC = epB + ep_n;
        int kv_pos = 0, rpos = 0;
        if (EPI == 2) { kv_pos = p.rope.state->pos0; rpos = p.rope.rope_pos ? p.rope.rope_pos[0] : kv_pos; }
        if (PRO == 1) { // the sum-of-squares exchange first
            if (lane == 0) red[wave] = 0.0;
            __syncthreads();
        }
        if (EPI == 1) {
            if (lane < PS_EXP2F_N) exp_tab[lane] = ps_exp2f_tab[lane];
        } else {
            for (int tl0 = 0; tl0 < nt; tl0 += 8) { // lane (r, u): row r of local task tl0 + u
                const int tl = tl0 + u;
                if (tl >= nt) continue;
                int wi = 0, grp = t0 + tl;
                if (MULTI && p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
                if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
                const float *b   = wi == 0 ? p.w[0].bias : (wi == 1 ? p.w[1].bias : p.w[2].bias);
                const int64_t row = (int64_t)grp * 8 + r; // (N % 8 == 0: every row exists)
                float va = 0.f, vb = 0.f, vc = 0.f;
                if (b) vc = b[row];
                if (EPI == 0) {
                    if (p.residual && wi == 0) va = p.residual[row];
                } else if (wi + p.rope_wi0 != 2) { // (cos, sin) of the rotation pair this row belongs to
                    const int e = (int)(row % p.rope.head_size);
                    if (e < p.rope.n_dims) {
                        const int64_t i0 = (int64_t)rpos * p.rope.head_size + (e & ~1);
                        va = p.rope.rope_table[i0]; vb = p.rope.rope_table[i0 + 1];
                    }
                }
                epA[tl * 8 + r] = va; epB[tl * 8 + r] = vb; epC[tl * 8 + r] = vc;
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(3); // one wave serves NW producers: it gets the issue slots first
        float acc = 0.f, summs = 0.f, ygate = 0.f;
        int tl = 0, un = 0; // local task, units of it already chained
        auto reduce = [&]() { // hsum_float_8 (ggml-quants.c:62-68) [+ summs]; valid in the lane with u == 0
            float v = __fadd_rn(acc, dpp_f<0x104>(acc));
            v = __fadd_rn(v, dpp_f<0x102>(v));
            v = __fadd_rn(v, dpp_f<0x101>(v));
            if (WT == PS_Q5_K) v = __fadd_rn(v, summs);
            acc = 0.f; summs = 0.f;
            return v;
        };
        auto row_done = [&]() {
            const float y = reduce();
            int wi = 0, grp = t0 + tl;
            if (EPI != 1) {
                if (MULTI && p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
                if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
            }
            float *o = p.w[0].out;
            const float *b = p.w[0].bias;
            if (wi == 1) { o = p.w[1].out; b = p.w[1].bias; }
            if (wi == 2) { o = p.w[2].out; b = p.w[2].bias; }
            const int64_t row = (int64_t)grp * 8 + r;
            const float ea = EPI != 1 ? epA[tl * 8 + r] : 0.f, eb = EPI == 2 ? epB[tl * 8 + r] : 0.f, ec = EPI != 1 ? epC[tl * 8 + r] : 0.f;
            if constexpr (EPI == 2) { // q / k: rotate adjacent pairs (rows 2i, 2i+1 sit in neighbouring lane groups); v: transpose-append
                float v = y;
                if (b) v = __fadd_rn(v, ec);
                const float vp = dpp_f<0x128>(v); // partner row (row_ror:8 swaps the two row groups of 8 lanes)
                const psk_rope_kv &R = p.rope;
                const int role = wi + p.rope_wi0; // 0 q, 1 k, 2 v
                if (u == 0) {
                    if (role == 2) {
                        R.v_cache[row * R.n_ctx + kv_pos] = v;
                        if (R.v16) R.v16[(int64_t)kv_pos * R.kv_dim + row] = (_Float16)v;
                    } else {
                        const int e = (int)(row % R.head_size);
                        float res = v;
                        if (e < R.n_dims) {
                            const float c = ea, sn = eb;
                            const float x0 = (e & 1) ? vp : v, x1 = (e & 1) ? v : vp;
                            res = ps_rope_one(x0, x1, c, sn, (e & 1) != 0);
                        }
                        if (role == 0) o[row] = res; else { R.k_cache[(int64_t)kv_pos * R.kv_dim + row] = res; if (R.k16) R.k16[(int64_t)kv_pos * R.kv_dim + row] = (_Float16)res; }
                    }
                }
            } else if (u == 0) {
                if (EPI == 1) {
                    ps_out_wt(o + row, g4_silu_mul(ygate, y, exp_tab));
                } else {
                    float v = y;
                    if (b) v = __fadd_rn(v, ec);
                    if (p.residual && wi == 0) v = __fadd_rn(ea, v);
                    ps_out_wt(o + row, v);
                }
            }
            un = 0;
            tl++;
        };
        auto batch = [&](auto nconst, const Rec *rb, const int k0) { // N records in one LDS round trip, then the chain
            constexpr int N = decltype(nconst)::value;
            Rec rc[N];
#pragma unroll
            for (int k = 0; k < N; k++) rc[k] = rb[(k0 + k) * 64];
#pragma unroll
            for (int k = 0; k < N; k++) {
                acc = __fmaf_rn(rc[k].x, rc[k].y, acc);
                if constexpr (WT == PS_Q5_K) summs = __fadd_rn(summs, rc[k].z);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int c = 0; c < DC * n_iters; c++) {
            __syncthreads();
            if (c >= n_chunks) continue;
            const Rec *rb  = recs + (size_t)(c & 1) * UPB * 64 + lane;
            const int kend = min(UPB, s_end - c * UPB);
            for (int k0 = 0; k0 < kend;) { // runs: units of one row (EPI 1: of one half of a gate/up pair)
                const int bound = (EPI == 1 && un < n_units) ? n_units : tot;
                const int len   = min(bound - un, kend - k0);
                int kk = k0, rem = len;
                for (; rem >= 8; rem -= 8, kk += 8) batch(std::integral_constant<int, 8>{}, rb, kk);
                if (rem >= 4) { batch(std::integral_constant<int, 4>{}, rb, kk); rem -= 4; kk += 4; }
                if (rem >= 2) { batch(std::integral_constant<int, 2>{}, rb, kk); rem -= 2; kk += 2; }
                if (rem >= 1) batch(std::integral_constant<int, 1>{}, rb, kk);
                un += len;
                k0 += len;
                if (EPI == 1 && un == n_units) ygate = reduce(); // gate row finished: reduce it, the chains restart for the up row
                if (un == tot) row_done();
            }
        }
    }
}

template <int WT, int NW, int UPW, int DC, int EA, int TPW, int EPI, int PRO>
void launch_gk(hipStream_t st, int grid, const GKParams &p) {
    const size_t smem = (size_t)p.act_bytes + (size_t)2 * NW * UPW * 64 * sizeof(typename GKRec<WT>::T) + (size_t)3 * (p.split_q + 1) * 8 * sizeof(float);
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr) && smem > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)gemvk_kernel<WT, NW, UPW, DC, EA, TPW, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    }
    psk_note_kernel("gemvk_kernel<%d, %d, %d, %d, %d, %d, %d, %d>", WT, NW, UPW, DC, EA, TPW, EPI, PRO);
    hipLaunchKernelGGL((gemvk_kernel<WT, NW, UPW, DC, EA, TPW, EPI, PRO>), dim3((unsigned)grid), dim3((NW + 1) * 64), smem, st, p);
}

template <int WT, int UPW, int DC, int EA, int TPW>
int launch_gk_ep(hipStream_t st, int grid, const GKParams &p, int epi, int pro) {
    constexpr int NW = 8;
    if (epi == 2) { if (pro != 1) return -1; launch_gk<WT, NW, UPW, DC, EA, TPW, 2, 1>(st, grid, p); return 0; }
    if constexpr (WT == PS_Q5_K) {
        if (epi == 1) { if (pro != 1) return -1; launch_gk<WT, NW, UPW, DC, EA, TPW, 1, 1>(st, grid, p); return 0; }
    } else if (epi != 0) return -1;
    if (pro == 0) return -1; // (pre-quantized single columns do not occur on the decode path: not instantiated)
    if (pro == 1) launch_gk<WT, NW, UPW, DC, EA, TPW, 0, 1>(st, grid, p);
    else launch_gk<WT, NW, UPW, DC, EA, TPW, 0, 2>(st, grid, p);
    return 0;
}
template <int WT, int UPW, int DC, int EA>
int launch_gk_kc(hipStream_t st, int grid, const GKParams &p, int epi, int pro) {
    if (p.n_units <= 16) return launch_gk_ep<WT, UPW, DC, EA, 2>(st, grid, p, epi, pro);
    if (p.n_units <= 64) return launch_gk_ep<WT, UPW, DC, EA, 8>(st, grid, p, epi, pro);
    return -1;
}
template <int WT>
int launch_gk_wt(hipStream_t st, int grid, const GKParams &p, int epi, int pro) {
    static const int cfg = getenv("PS_GEMVK_CFG") ? atoi(getenv("PS_GEMVK_CFG")) : 0; // (ring shape, for measurements)
    switch (cfg) {
    case 1: return launch_gk_kc<WT, 3, 2, 1>(st, grid, p, epi, pro); // 8B Q5_K_M 390 tok/s against 413 with the default, Q4_K_M 452 / 457
    case 2: return launch_gk_kc<WT, 2, 4, 1>(st, grid, p, epi, pro);
    case 3: return launch_gk_kc<WT, 2, 3, 2>(st, grid, p, epi, pro);
    case 4: return launch_gk_kc<WT, 1, 6, 2>(st, grid, p, epi, pro);
    default: return launch_gk_kc<WT, 2, 3, 1>(st, grid, p, epi, pro);
    }
}

} // namespace

bool psk_gemvk_covers(int wt, int64_t K) {
    static const bool off = getenv("PS_NO_GEMVK") != nullptr; // (A/B switch for measurements)
    return !off && (wt == PS_Q6_K || wt == PS_Q5_K) && K >= 256 && K % 256 == 0 && K <= 16384;
}

// Single-column Q6_K / Q5_K mat-vec; all matrices of one type.  Returns -1 when the launch is not covered (the caller falls back
// to a quantizer launch + k_gemv6.hip).
int psk_gemvk(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K) {
    if (a.n_w < 1 || a.n_w > 3) return -1;
    const int wt = a.w[0]->dtype;
    if (!psk_gemvk_covers(wt, K)) return -1;
    GKParams p{};
    int groups_total = 0;
    for (int i = 0; i < a.n_w; i++) {
        const ps_weight *w = a.w[i];
        if (w->dtype != wt || w->K != K || w->N % 8) return -1;
        const int ng = (int)(w->N / 8);
        p.w[i] = GKMat{w->qs, w->qh, w->sc, (const uint16_t *)w->aux, a.out[i], a.bias[i], w->N, ng};
        groups_total += ng;
    }
    const int epi = a.silu_pair ? 1 : (a.rope ? 2 : 0);
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N || a.pro != 1)) return -1;
    if (epi == 2) {
        if (a.n_w + a.rope_wi0 > 3 || a.rope_wi0 < 0 || a.pro != 1) return -1;
        p.rope = *a.rope; p.rope_wi0 = a.rope_wi0;
    }
    if (wt == PS_Q6_K && (epi == 1 || a.n_w != 1)) return -1;
    p.n_w = a.n_w; p.n_units = (int)(K / 256); p.K = (int)K;
    p.n_tasks = epi == 1 ? p.w[0].n_groups : groups_total;
    p.residual = a.residual; p.x = a.pro_x; p.nw = a.pro_norm_w; p.eps = a.pro_eps;
    p.aq = act.qs; p.ad = act.d; p.abs16 = act.bs16;
    p.act_bytes = (int)((K + (K / 256) * 4 + (K / 32) * 4 + 15) / 16 * 16);
    int grid = p.n_tasks < n_cu ? p.n_tasks : n_cu;
    if (grid < 1) return -1;
    p.split_q = p.n_tasks / grid; p.split_r = p.n_tasks % grid;
    return wt == PS_Q6_K ? launch_gk_wt<PS_Q6_K>(st, grid, p, epi, a.pro) : launch_gk_wt<PS_Q5_K>(st, grid, p, epi, a.pro);
}
