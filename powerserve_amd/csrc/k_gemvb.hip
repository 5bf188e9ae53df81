// Decode mat-vec for the 32-element block formats (Q4_0 / Q8_0 weights x one Q8_0 activation column): the producer /
// chain-wave form of gemv4 (k_gemv4.hip) for ggml_vec_dot_q4_0_q8_0 and ggml_vec_dot_q8_0_q8_0
// (libs/ggml/src/ggml-quants.c:4205-4228, :5761-5782), bit-exact: per block  acc[u] = fma(d_w * d_y, (float)sumi[u], acc[u])
// in block order, then hsum_float_8 (:62-68).  What the 1B / 0.5B shapes showed (profiles/r03_decode_kernel_stats_1b_q4_0.txt:
// down 12.3 us for 9.4 MB, QKV 10.7 us for 3.5 MB behind 1024-thread workgroups whose chain waves also derived every block
// scale) is fixed the way gemv4 fixed Q4_K:
//   * 576-thread workgroups: eight producer waves + ONE chain wave (a kernel boundary costs ~1.5 us behind 512-thread
//     workgroups, ~3 us behind 1024-thread ones, profiles/r02_micro_boundary.txt);
//   * a producer turns a 1 KiB unit (4 blocks of 8 / 16 rows) into ready fp32 records -- per block {d_w * d_y, (float)sumi}
//     (Q4_0: the low- and the high-nibble sum) -- so the chain wave's share is one fma per block and chain and nothing else;
//   * the activation row is requested first and quantized once per workgroup into LDS in a unit-major order (dword
//     [unit][quad][block]) so that a lane fetches the four blocks' quads of a unit with one ds_read_b128.  The workgroup's
//     one LDS pipe is what the long streams run against (a b128 access of a wave occupies it for 8 clocks: y, scales, records
//     written and read are 9 of them per unit), so Q4_0's  sum (q - 8) y  takes no correction term from LDS: a nibble moved to
//     the high half of its byte and xor 0x80 IS the signed byte 16 (q - 8); v_dot4 returns 16 sumi and the activation scale
//     is stored as d_y / 16 (powers of two commute with the rounding of d_w * d_y and of the fma: same bits);
//   * a register ring of four 16 KiB chunks per workgroup (two units per producer and chunk): 64 KiB in flight per CU, the
//     first two chunks requested before the prologue, the others behind the quantizer;
//   * any row length that is a multiple of 128 (Qwen2-0.5B: 7 and 38 units per row): a unit of the stream is placed on its
//     own (task, unit), rows may end anywhere inside a chunk.
// Epilogues as gemv4: EPI 0 bias / residual, EPI 1 SiLU(gate) * up, EPI 2 adjacent-pair RoPE + KV-cache append (Q / K / V).
#include "ps_gemv_dev.h"

namespace {

template <int WT> struct GBT;
template <> struct GBT<PS_Q4_0> { static constexpr int RG = 16, LPR = 4, RECF = 3; }; // rows per group, lanes per row, float4s per record
template <> struct GBT<PS_Q8_0> { static constexpr int RG = 8, LPR = 8, RECF = 2; };

__device__ __forceinline__ float gb_silu_mul(float g, float u, const uint64_t *tab) { // src/backend/ggml/ggml.cpp:115-129, expf table in LDS
    float val = g;
    val       = __fmul_rn(val, __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-val, tab))));
    return __fmul_rn(val, u);
}

struct GBMat {
    const uint8_t *qs, *aux;
    float *out;
    const float *bias;
    int64_t N;
    int n_groups;
};
struct GBParams {
    GBMat w[3];
    int n_w, n_units, n_tasks;  // tasks: row groups (EPI 0 / 2) or gate/up row-group pairs (EPI 1)
    int split_q, split_r;       // tasks per workgroup = split_q (+1 for the first split_r workgroups)
    int K, act_bytes;
    const float *residual;
    const float *x, *nw;        // PRO 1: rmsnorm(x, nw, eps) then quantize;  PRO 2: quantize(x)
    float eps;
    const int8_t *aq;           // PRO 0: activation already quantized (natural order)
    const float *ad;
    psk_rope_kv rope;           // EPI 2
};

// dword (block, quad) of the activation row in LDS: [unit = block / 4][quad][block % 4]
__device__ __forceinline__ int gb_ydw(const int blk, const int quad) { return ((blk >> 2) << 5) + (quad << 2) + (blk & 3); }

// quantize_row_q8_0 (AVX2 branch, ggml-quants.c:957-1039) of the lane's four values e .. e+3 (a block = 8 consecutive lanes)
template <int WT>
__device__ __forceinline__ void gb_quantize(const float v[4], const int e, const bool live, int *lq, float *ld) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax       = group8_max_dpp(amax);
    const float dd = __fdiv_rn(amax, 127.f);
    const float id = (amax != 0.0f) ? __fdiv_rn(127.f, amax) : 0.0f;
    int q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = __float2int_rn(__fmul_rn(v[i], id)); // round-half-even
    if (live) {
        const int blk = e >> 5, quad = (e >> 2) & 7, idx = gb_ydw(blk, quad);
        lq[idx] = (int)((uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24));
        if (quad == 0) ld[blk] = (WT == PS_Q4_0 ? 0.0625f : 1.0f) * ps_h2f(ps_f2h(dd));
    }
}

// NW producer waves, TPW activation tiles (256 elements) per producer
template <int WT, int NW, int TPW, int EPI, int PRO>
__global__ __launch_bounds__((NW + 1) * 64) void gemvb_kernel(const GBParams p) {
    using T = GBT<WT>;
    constexpr int UPW = 2, DC = 4, UPB = NW * UPW, RG = T::RG, LPR = T::LPR, RECF = T::RECF;
    constexpr int AUXU = RG * 8; // header bytes per unit: four fp16 block scales per row
    constexpr uint32_t M = 0x0F0F0F0Fu;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ uint64_t exp_tab[PS_EXP2F_N];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K, n_units = p.n_units;
    int *lq      = (int *)smem;           // [K / 4] quants, unit-major (gb_ydw)
    float *ld    = (float *)(lq + K / 4); // [K / 32] block scales (Q4_0: d_y / 16, see above)
    float4 *recs = (float4 *)(smem + p.act_bytes);          // [2][UPB][RECF][64]
    float *epA   = (float *)(recs + 2 * UPB * RECF * 64);   // [3][ep_n]: epilogue operands of this workgroup's rows
    const int r = lane / LPR, u = lane % LPR;

    const int tot = (EPI == 1) ? 2 * n_units : n_units; // stream units per task (EPI 1: gate units then up units)
    const int t0  = (int)blockIdx.x * p.split_q + min((int)blockIdx.x, p.split_r);
    const int nt  = p.split_q + ((int)blockIdx.x < p.split_r ? 1 : 0);
    const int s_end    = nt * tot;
    const int n_chunks = (s_end + UPB - 1) / UPB;
    const int n_iters  = (n_chunks + DC - 1) / DC;

    if (wave < NW) { // ------------------------------------------------------------------ producers
        float4 xv[TPW], wv[TPW];
        if (PRO != 0) ps_qrow_load<(PRO == 1 ? 1 : 0), TPW>(p.x, p.nw, K, xv, wv, NW);
        // every unit of the ring has its own place (local task, unit inside the task) in the stream; a trip moves it DC chunks on
        int step_t = 0, step_u = DC * UPB;
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
        ps_u32x4 q[DC][UPW];
        ps_u32x2 h[DC][UPW];
        const uint32_t lane16 = (uint32_t)lane * 16u, raux = (uint32_t)r * 8u;
        // loads are UNCONDITIONAL (a unit past the range re-reads 16 bytes of the first matrix) so that the compiler counts
        // vmcnt exactly and a chunk is consumed while the next ones are in flight
        auto issue = [&](const int d) {
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                const bool live = tS[d][i] < nt;
                int grp = t0 + (live ? tS[d][i] : 0), ul = live ? uS[d][i] : 0;
                const uint8_t *qb = p.w[0].qs, *ab = p.w[0].aux;
                if (EPI == 1) {
                    if (ul >= n_units) { ul -= n_units; qb = p.w[1].qs; ab = p.w[1].aux; }
                } else if (p.n_w > 1 && grp >= p.w[0].n_groups) {
                    grp -= p.w[0].n_groups; qb = p.w[1].qs; ab = p.w[1].aux;
                    if (p.n_w > 2 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; qb = p.w[2].qs; ab = p.w[2].aux; }
                }
                const uint32_t idx = (uint32_t)(grp * n_units + ul);
                q[d][i] = __builtin_nontemporal_load((const ps_u32x4 *)(qb + ((uint64_t)idx << 10) + (live ? lane16 : 0u)));
                h[d][i] = *(const ps_u32x2 *)(ab + (uint64_t)idx * AUXU + (live ? raux : 0u));
            }
        };
        auto produce = [&](const int d, const int buf) {
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                if (tS[d][i] >= nt) continue; // wave-uniform: past this workgroup's range
                const int un = uS[d][i], ul = (EPI == 1 && un >= n_units) ? un - n_units : un;
                const uint32_t wq[4] = {q[d][i].x, q[d][i].y, q[d][i].z, q[d][i].w};
                const float4 yd = *(const float4 *)(ld + ul * 4);
                const float dd[4] = {__fmul_rn(ps_h2f((uint16_t)(h[d][i].x & 0xffff)), yd.x), __fmul_rn(ps_h2f((uint16_t)(h[d][i].x >> 16)), yd.y),
                                     __fmul_rn(ps_h2f((uint16_t)(h[d][i].y & 0xffff)), yd.z), __fmul_rn(ps_h2f((uint16_t)(h[d][i].y >> 16)), yd.w)};
                float4 *out = recs + (size_t)((buf * UPB + wave * UPW + i) * RECF) * 64 + lane;
                if constexpr (WT == PS_Q4_0) { // lane u' holds bytes 4u' .. 4u'+3 of each block: low nibbles = quad u', high = quad u' + 4
                    const int4 yl = *(const int4 *)(lq + ul * 32 + u * 4), yh = *(const int4 *)(lq + ul * 32 + (u + 4) * 4);
                    int tl[4], th[4]; // bytes 16 (q - 8): sixteen times the reference's sums (dd carries the 1 / 16)
                    dot4x4(tl, (int)(((wq[0] << 4) & ~M) ^ 0x80808080u), (int)(((wq[1] << 4) & ~M) ^ 0x80808080u), (int)(((wq[2] << 4) & ~M) ^ 0x80808080u),
                           (int)(((wq[3] << 4) & ~M) ^ 0x80808080u), yl.x, yl.y, yl.z, yl.w);
                    dot4x4(th, (int)((wq[0] & ~M) ^ 0x80808080u), (int)((wq[1] & ~M) ^ 0x80808080u), (int)((wq[2] & ~M) ^ 0x80808080u),
                           (int)((wq[3] & ~M) ^ 0x80808080u), yh.x, yh.y, yh.z, yh.w);
                    float sl[4], sh[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) { sl[b] = (float)tl[b]; sh[b] = (float)th[b]; }
                    out[0]   = make_float4(dd[0], sl[0], sh[0], dd[1]);
                    out[64]  = make_float4(sl[1], sh[1], dd[2], sl[2]);
                    out[128] = make_float4(sh[2], dd[3], sl[3], sh[3]);
                } else {
                    const int4 y = *(const int4 *)(lq + ul * 32 + u * 4);
                    int s[4];
                    dot4x4(s, (int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3], y.x, y.y, y.z, y.w);
                    out[0]  = make_float4(dd[0], (float)s[0], dd[1], (float)s[1]);
                    out[64] = make_float4(dd[2], (float)s[2], dd[3], (float)s[3]);
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
        // the CU's vector-memory queue is served in order: the activation row first, two chunks (32 KiB per CU) behind it, the
        // other two once the quantizer has run (a wave whose requests do not fit stalls IN the issue, on the prologue's path)
        constexpr int EARLY = PRO == 0 ? DC : 2;
#pragma unroll
        for (int d = 0; d < EARLY; d++) issue(d);
        if (PRO == 0) {
            for (int i = threadIdx.x; i < K / 4; i += NW * 64) {
                const int y = ((const int *)p.aq)[i], idx = gb_ydw(i >> 3, i & 7);
                lq[idx] = y;
            }
            for (int i = threadIdx.x; i < K / 32; i += NW * 64) ld[i] = (WT == PS_Q4_0 ? 0.0625f : 1.0f) * p.ad[i];
        } else {
            // RMSNorm (PRO 1: ggml.c:12667-12720, double sum of squares, scale = 1/sqrtf(mean + eps), y = x * (w * scale)), then
            // the Q8_0 quantizer over this wave's tiles (tile t = wave + i * NW)
            float scale = 1.0f;
            if (PRO == 1) {
                double ss = 0.0;
#pragma unroll
                for (int i = 0; i < TPW; i++) {
                    ss += (double)__fmul_rn(xv[i].x, xv[i].x);
                    ss += (double)__fmul_rn(xv[i].y, xv[i].y);
                    ss += (double)__fmul_rn(xv[i].z, xv[i].z);
                    ss += (double)__fmul_rn(xv[i].w, xv[i].w);
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
            for (int i = 0; i < TPW; i++) {
                const int e = (wave + i * NW) * 256 + lane * 4;
                float v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}; // (zeros past the row's end)
                if (PRO == 1) {
                    v[0] = __fmul_rn(v[0], __fmul_rn(wv[i].x, scale));
                    v[1] = __fmul_rn(v[1], __fmul_rn(wv[i].y, scale));
                    v[2] = __fmul_rn(v[2], __fmul_rn(wv[i].z, scale));
                    v[3] = __fmul_rn(v[3], __fmul_rn(wv[i].w, scale));
                }
                gb_quantize<WT>(v, e, e < K, lq, ld);
            }
#pragma unroll
            for (int d = EARLY; d < DC; d++) issue(d);
        }
        __syncthreads();
        // chunk it * DC + d from ring slot d
        for (int it = 0; it < n_iters; it++) {
#pragma unroll
            for (int d = 0; d < DC; d++) {
                produce(d, d & 1); // (DC is even: chunk it * DC + d has the parity of d)
                advance(d);
                issue(d);
                __syncthreads();
            }
        }
    } else { // ------------------------------------------------------------------------- chain wave
        // everything the epilogue reads from memory is fetched now, into LDS, while this wave has nothing to chain
        const int ep_n = (p.split_q + 1) * RG;
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
            constexpr int TP = 64 / RG; // tasks per pass: lane = (task tl0 + lane / RG, row lane % RG)
            const int rr = lane % RG;
            for (int tl0 = 0; tl0 < nt; tl0 += TP) {
                const int tl = tl0 + lane / RG;
                if (tl >= nt) continue;
                int wi = 0, grp = t0 + tl;
                if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
                if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
                const int64_t Nw = wi == 0 ? p.w[0].N : (wi == 1 ? p.w[1].N : p.w[2].N);
                const float *b   = wi == 0 ? p.w[0].bias : (wi == 1 ? p.w[1].bias : p.w[2].bias);
                const int64_t row = (int64_t)grp * RG + rr;
                float va = 0.f, vb = 0.f, vc = 0.f;
                if (row < Nw) {
                    if (b) vc = b[row];
                    if (EPI == 0) {
                        if (p.residual && wi == 0) va = p.residual[row];
                    } else if (wi != 2) { // (cos, sin) of the rotation pair this row belongs to
                        const int e = (int)(row % p.rope.head_size);
                        if (e < p.rope.n_dims) {
                            const int64_t i0 = (int64_t)rpos * p.rope.head_size + (e & ~1);
                            va = p.rope.rope_table[i0]; vb = p.rope.rope_table[i0 + 1];
                        }
                    }
                }
                epA[tl * RG + rr] = va; epB[tl * RG + rr] = vb; epC[tl * RG + rr] = vc;
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(3); // one wave serves NW producers: it gets the issue slots first
        float acc0 = 0.f, acc1 = 0.f, ygate = 0.f;
        int tl = 0, un = 0; // local task, units of it already chained
        auto row_done = [&]() {
            const float y = row_reduce<WT>(acc0, acc1, 0.f);
            int wi = 0, grp = t0 + tl;
            if (EPI != 1) {
                if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
                if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
            }
            int64_t Nw = p.w[0].N;
            float *o = p.w[0].out;
            const float *b = p.w[0].bias;
            if (wi == 1) { Nw = p.w[1].N; o = p.w[1].out; b = p.w[1].bias; }
            if (wi == 2) { Nw = p.w[2].N; o = p.w[2].out; b = p.w[2].bias; }
            const int64_t row = (int64_t)grp * RG + r;
            const float ea = EPI != 1 ? epA[tl * RG + r] : 0.f, eb = EPI == 2 ? epB[tl * RG + r] : 0.f, ec = EPI != 1 ? epC[tl * RG + r] : 0.f;
            if constexpr (EPI == 2) { // q / k: rotate adjacent pairs (rows 2i, 2i+1 sit in neighbouring lane groups); v: transpose-append
                float v = y;
                if (b && row < Nw) v = __fadd_rn(v, ec);
                const float vp = (LPR == 8) ? dpp_f<0x128>(v) : __shfl_xor(v, LPR, 64); // partner row
                const psk_rope_kv &R = p.rope;
                if (u == 0 && row < Nw) {
                    if (wi == 2) {
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
                        if (wi == 0) o[row] = res; else { R.k_cache[(int64_t)kv_pos * R.kv_dim + row] = res; if (R.k16) R.k16[(int64_t)kv_pos * R.kv_dim + row] = (_Float16)res; }
                    }
                }
            } else if (u == 0 && row < Nw) {
                if (EPI == 1) {
                    ps_out_wt(o + row, gb_silu_mul(ygate, y, exp_tab));
                } else {
                    float v = y;
                    if (b) v = __fadd_rn(v, ec);
                    if (p.residual && wi == 0) v = __fadd_rn(ea, v);
                    ps_out_wt(o + row, v);
                }
            }
            acc0 = 0.f; acc1 = 0.f;
            un = 0;
            tl++;
        };
        auto batch = [&](auto nconst, const float4 *rb, const int k0) { // N records in one LDS round trip, then the fma chains
            constexpr int N = decltype(nconst)::value;
            float4 rc[N][RECF];
#pragma unroll
            for (int k = 0; k < N; k++)
#pragma unroll
                for (int j = 0; j < RECF; j++) rc[k][j] = rb[((k0 + k) * RECF + j) * 64];
#pragma unroll
            for (int k = 0; k < N; k++) {
                if constexpr (WT == PS_Q4_0) {
                    const float4 a = rc[k][0], b = rc[k][1], c = rc[k][2];
                    acc0 = __fmaf_rn(a.x, a.y, acc0); acc1 = __fmaf_rn(a.x, a.z, acc1);
                    acc0 = __fmaf_rn(a.w, b.x, acc0); acc1 = __fmaf_rn(a.w, b.y, acc1);
                    acc0 = __fmaf_rn(b.z, b.w, acc0); acc1 = __fmaf_rn(b.z, c.x, acc1);
                    acc0 = __fmaf_rn(c.y, c.z, acc0); acc1 = __fmaf_rn(c.y, c.w, acc1);
                } else {
                    const float4 a = rc[k][0], b = rc[k][1];
                    acc0 = __fmaf_rn(a.x, a.y, acc0); acc0 = __fmaf_rn(a.z, a.w, acc0);
                    acc0 = __fmaf_rn(b.x, b.y, acc0); acc0 = __fmaf_rn(b.z, b.w, acc0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int c = 0; c < DC * n_iters; c++) {
            __syncthreads();
            if (c >= n_chunks) continue;
            const float4 *rb = recs + (size_t)(c & 1) * UPB * RECF * 64 + lane;
            const int kend   = min(UPB, s_end - c * UPB);
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
                if (EPI == 1 && un == n_units) { // gate row finished: reduce it, restart the chains for the up row
                    ygate = row_reduce<WT>(acc0, acc1, 0.f);
                    acc0 = 0.f; acc1 = 0.f;
                }
                if (un == tot) row_done();
            }
        }
    }
}

template <int WT, int NW, int TPW, int EPI, int PRO>
void launch_gb(hipStream_t st, int grid, const GBParams &p) {
    constexpr int UPB = NW * 2, RECF = GBT<WT>::RECF, RG = GBT<WT>::RG;
    const size_t smem = (size_t)p.act_bytes + (size_t)2 * UPB * RECF * 64 * sizeof(float4) + (size_t)3 * (p.split_q + 1) * RG * sizeof(float);
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) {
        (void)hipFuncSetAttribute((const void *)gemvb_kernel<WT, NW, TPW, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
    psk_note_kernel("gemvb_kernel<%d, %d, %d, %d, %d>", WT, NW, TPW, EPI, PRO);
    hipLaunchKernelGGL((gemvb_kernel<WT, NW, TPW, EPI, PRO>), dim3((unsigned)grid), dim3((NW + 1) * 64), smem, st, p);
}

template <int WT, int TPW>
int launch_gb_ep(hipStream_t st, int grid, const GBParams &p, int epi, int pro) {
    constexpr int NW = 8;
    if (epi == 2) { if (pro != 1) return -1; launch_gb<WT, NW, TPW, 2, 1>(st, grid, p); return 0; }
    if (epi == 1) { if (pro != 1) return -1; launch_gb<WT, NW, TPW, 1, 1>(st, grid, p); return 0; }
    if (pro == 0) launch_gb<WT, NW, TPW, 0, 0>(st, grid, p);
    else if (pro == 1) launch_gb<WT, NW, TPW, 0, 1>(st, grid, p);
    else launch_gb<WT, NW, TPW, 0, 2>(st, grid, p);
    return 0;
}
template <int WT>
int launch_gb_wt(hipStream_t st, int grid, const GBParams &p, int epi, int pro) {
    const int tiles = (p.K + 255) / 256;
    if (tiles <= 8) return launch_gb_ep<WT, 1>(st, grid, p, epi, pro);
    if (tiles <= 16) return launch_gb_ep<WT, 2>(st, grid, p, epi, pro);
    if (tiles <= 32) return launch_gb_ep<WT, 4>(st, grid, p, epi, pro);
    if (tiles <= 64) return launch_gb_ep<WT, 8>(st, grid, p, epi, pro);
    return -1;
}

} // namespace

// row lengths / weight types the kernel takes (psk_gemv_rope_ok asks before the model plans a fused QKV launch)
bool psk_gemvb_covers(int wt, int64_t K) {
    static const bool off = getenv("PS_NO_GEMVB") != nullptr; // (A/B switch for measurements)
    return !off && (wt == PS_Q4_0 || wt == PS_Q8_0) && K >= 128 && K % 128 == 0 && K <= 16384;
}

// Single-column Q4_0 / Q8_0 mat-vec.  Returns -1 when the launch is not covered (the caller falls back to gemv1 / gemv_kernel).
int psk_gemvb(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K) {
    if (a.n_w < 1 || a.n_w > 3) return -1;
    const int wt = a.w[0]->dtype;
    if (!psk_gemvb_covers(wt, K)) return -1;
    const int rg = wt == PS_Q4_0 ? 16 : 8;
    GBParams p{};
    int groups_total = 0;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != wt || a.w[i]->K != K) return -1;
        const int ng = (int)((a.w[i]->N + rg - 1) / rg);
        p.w[i] = GBMat{a.w[i]->qs, a.w[i]->aux, a.out[i], a.bias[i], a.w[i]->N, ng};
        groups_total += ng;
    }
    const int epi = a.silu_pair ? 1 : (a.rope ? 2 : 0);
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N || a.pro != 1)) return -1;
    if (epi == 2) {
        if (a.n_w != 3 || a.pro != 1) return -1;
        p.rope = *a.rope;
    }
    p.n_w = a.n_w; p.n_units = (int)(K / 128); p.K = (int)K;
    p.n_tasks = epi == 1 ? p.w[0].n_groups : groups_total;
    p.residual = a.residual; p.x = a.pro_x; p.nw = a.pro_norm_w; p.eps = a.pro_eps;
    p.aq = act.qs; p.ad = act.d;
    p.act_bytes = (int)((K + K / 8 + 15) / 16 * 16);
    int grid = p.n_tasks < n_cu ? p.n_tasks : n_cu;
    if (grid < 1) return -1;
    p.split_q = p.n_tasks / grid; p.split_r = p.n_tasks % grid;
    // Measured per launch (profiles/r03_decode_kernel_stats_{1b_q4_0,05b_q8_0}*.txt), this kernel against the register-resident
    // gemv1 (k_gemv.hip), us: rows of 16+ units 1B QKV 6.8 / 10.7, O 5.7 / 6.2, gate/up 8.7 / 11.7, down 8.0 / 12.3; 0.5B gate/up
    // (2 x 7 units) 7.5 / 7.8, down (38) 6.5 / 7.9 -- but rows of 7 units (0.5B QKV 5.4 / 4.6, O 5.7 / 5.4: half a chunk per
    // workgroup, gemv1's 256-thread workgroups start faster) and the long single-matrix stream of the lm_head (1B 40.5 / 37.8,
    // 0.5B 48 / 41: the one LDS pipe of a workgroup carries three record float4s per unit both ways) go to gemv1.
    const int tot = epi == 1 ? 2 * p.n_units : p.n_units;
    if (epi != 2 && tot <= 8) return -1;
    if (epi == 0 && a.n_w == 1 && p.split_q >= 8 && tot <= 64) return -1;
    const size_t smem = (size_t)p.act_bytes + (size_t)2 * 16 * 3 * 64 * 16 + (size_t)3 * (p.split_q + 1) * rg * 4;
    if (smem > 156 * 1024) return -1;
    return wt == PS_Q4_0 ? launch_gb_wt<PS_Q4_0>(st, grid, p, epi, a.pro) : launch_gb_wt<PS_Q8_0>(st, grid, p, epi, a.pro);
}
