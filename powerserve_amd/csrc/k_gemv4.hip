// Decode mat-vec, second generation (single activation column, Q4_K weights): the same bit-exact arithmetic as
// round 1's gemv3_kernel (deleted in round 3: producers turn 1 KiB units into the reference's integer partials, a chain wave runs the
// fp32 fma chains of ggml_vec_dot_q4_K_q8_K in unit order, libs/ggml/src/ggml-quants.c:7809-7873), restructured around
// what the launch-boundary micro-benchmark (tools/micro/overlap.hip, profiles/r02_micro_boundary.txt) showed on MI355X:
//   * a dependent kernel boundary costs 2.8-3.1 us behind 1024-thread workgroups but 1.3-1.6 us behind 512-thread ones,
//     and a 16.8 MB streaming launch takes 4.2-4.5 us with 8 waves per CU against 6.1 us with 16: 64 KiB of loads in
//     flight per CU already saturate HBM (6.2 TB/s), more only queue -> NW = 7 or 8 producer waves + ONE chain wave;
//   * the activation row is requested FIRST (vmcnt retires in order) and its tiles are dealt to all producers; the two
//     first chunks of weights go out right behind it, before any barrier;
//   * every address of a wave's four units of a chunk comes from ONE scalar computation (the four units are consecutive
//     1 KiB pieces of one row group: rows end on multiples of four units), so the inner loop carries a handful of SALU
//     instructions per chunk instead of ~60 per unit;
//   * chunks are sized so that whole rows fit (K = 4096: 8 producers, 32 units = 2 row groups; K = 14336: 7 producers,
//     28 units = half a row group).
// Epilogues: EPI 0 bias / residual, EPI 1 SiLU(gate)*up, EPI 2 RoPE + KV-cache append (QKV).
#include "ps_g4_dev.h"
#ifndef G4_OUT_WT
#define G4_OUT_WT 1 // (round 6) the result rows are stored write-through: nothing dirty is left for the end-of-kernel write-back, the next launch finds the row in memory; same-box A/B,
                    // three rounds (profiles/r06_out_wt_ab.txt): all mat-vecs of a token 1.239 -> 1.226 ms, 8B decode 596.3 -> 599.7 tok/s
#endif
namespace { __device__ __forceinline__ void g4_out(float *p, const float v) { if (G4_OUT_WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; } }

namespace {
constexpr int G4_PAIR = 0; // units of a chunk the scheduler may interleave (0: one at a time)

struct G4Mat {
    const uint8_t *qs, *aux;
    float *out;
    const float *bias;
    int64_t N;
    int n_groups;
};
struct G4Params {
    G4Mat w[3];
    int n_w, n_units, n_tasks;  // tasks: row groups (EPI 0 / 2) or gate/up row-group pairs (EPI 1)
    int split_q, split_r;       // tasks per workgroup = split_q (+1 for the first split_r workgroups)
    int K, col_bytes;
    const float *residual;
    const float *x, *nw;        // PRO 1: rmsnorm(x, nw, eps) then quantize;  PRO 2: quantize(x)
    float eps;
    const int8_t *aq;           // PRO 0: activation already quantized
    const float *ad;
    const int16_t *abs16;
    unsigned long long *dbg;
    psk_rope_kv rope;           // EPI 2
    int rope_wi0;               // EPI 2: w[0]'s place in the Q / K / V triple (a launch may carry a part of it)
};

// NW producer waves, DC chunks of weight loads in flight per wave (register ring), TPW activation tiles per producer,
// XW: wait for the activation row before the first weight request goes out (short launches: the row does not queue
// behind the whole chip's first burst of weights)
// YS: the activation operands of a wave's units stay in registers (1: the wave meets the same four super-blocks in every chunk,
// UPB % tot == 0; 2: two alternating sets, one per ring slot, 2 UPB % tot == 0 and DC == 2; 0: fetched from LDS per unit)
template <int NW, int DC, int TPW, int XW, int EPI, int PRO, int YS = 0>
__global__ __launch_bounds__((NW + 1) * 64) void gemv4_kernel(const G4Params p) {
    constexpr int WT = PS_Q4_K;
    using TR  = WTraits<WT>;
    // record of one (unit, lane): {d * yd, (float)sumi, -dmin * yd, (float)(mins . bsums)} -- everything up to the two
    // fmas of the chain is done by the producers, in parallel (a lone chain wave issues one instruction every ~4 cycles:
    // at 12 instructions per record it was the bottleneck of the launch, 1.3 us per 32-unit chunk)
    using Rec = float4;
    constexpr int UPW = 4, UPB = NW * UPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[16];
    __shared__ uint64_t exp_tab[PS_EXP2F_N];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K, n_units = p.n_units;
    const int nb32 = K / 32;
    int8_t *lq   = (int8_t *)smem;
    float *ld    = (float *)(smem + K);
    int *lb      = (int *)(ld + n_units);
    int16_t *l16 = (int16_t *)(lb + nb32);
    Rec *recs    = (Rec *)(smem + p.col_bytes); // [2][UPB][64]
    float *epA   = (float *)(recs + 2 * UPB * 64);  // [3][ep_cap * 8]: epilogue operands of this workgroup's rows
    LAct A;
    A.q32 = (const int *)lq; A.d = ld; A.bs32 = lb;
    const int r = lane >> 3, u = lane & 7;

    const int tot = (EPI == 1) ? 2 * n_units : n_units; // stream units per task (EPI 1: gate units then up units)
    const int t0  = (int)blockIdx.x * p.split_q + min((int)blockIdx.x, p.split_r);
    const int nt  = p.split_q + ((int)blockIdx.x < p.split_r ? 1 : 0);
    const int s_end    = nt * tot;                  // stream units of this workgroup
    const int n_chunks = (s_end + UPB - 1) / UPB;
    const int n_iters  = (n_chunks + DC - 1) / DC;
    unsigned long long *const pdbg = PS_TL(p.dbg); // (null unless the library is built with -DPS_TIMELINE=1: ps_dev.h)
    unsigned long long *const dbg = (pdbg && blockIdx.x < 1024 && lane == 0 && (wave == 0 || wave == NW)) ? pdbg + ((size_t)blockIdx.x * 2 + (wave == NW)) * 32 : nullptr;
    int dbg_n = 0;
    auto mark = [&]() { if (dbg && dbg_n < 28) dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); };
    mark(); // 0: entry
    if (dbg) dbg[29] = __builtin_amdgcn_s_memrealtime();
    if (pdbg && blockIdx.x < 1024 && lane == 0 && wave < 16) pdbg[((size_t)blockIdx.x * 2 + 1) * 32 + 12 + wave] = __builtin_amdgcn_s_memtime(); // every wave's entry

    if (wave < NW) { // ------------------------------------------------------------------ producers
        // 1. the activation row, dealt tile by tile to the producers (tile t -> wave t % NW)
        // (two tiles per wave-instruction, g4_quantize_pair: pair tp = wave + i * NW, lane l holds elements tp * 512 + 8 l .. + 7.  Measured,
        //  decode tok/s of the 8B: one tile per instruction 520, two 535, four 524 -- four leave half the waves without a tile at K = 4096)
        constexpr int PPW = (TPW + 1) / 2;
        const int n_pairs = n_units / 2;
        float4 xv[PPW][2], wv[PPW][2];
        if (PRO != 0) {
#pragma unroll
            for (int i = 0; i < PPW; i++) {
                const int tp = wave + i * NW;
                const int64_t e = (int64_t)(tp < n_pairs ? tp : 0) * 512 + lane * 8; // (never a branch around a load: a dead slot re-reads pair 0 and stores nothing)
                xv[i][0] = *(const float4 *)(p.x + e); xv[i][1] = *(const float4 *)(p.x + e + 4);
                if (PRO == 1) { wv[i][0] = *(const float4 *)(p.nw + e); wv[i][1] = *(const float4 *)(p.nw + e + 4); }
            }
        }
        // 2. this wave's slots of chunks 0 .. DC-1: (local task, unit inside the task); a slot is four consecutive units
        //    of one row group
        const int step_t = (DC * UPB) / tot, step_u = (DC * UPB) % tot; // a slot moves DC chunks per trip
        int tS[DC], uS[DC];
#pragma unroll
        for (int d = 0; d < DC; d++) {
            int t = 0, un = d * UPB + wave * UPW;
            while (un >= tot) { un -= tot; t++; }
            tS[d] = t; uS[d] = un;
        }
        // Ring of weight chunks in registers: per chunk and lane four 16-byte quant loads and ONE 16-byte header load — the
        // 4 x 128 B of a slot's headers are contiguous, lane l < 32 fetches piece l, and the wave hands them round through
        // 512 B of LDS when the chunk is produced (a header register per unit and lane made the ring 32 registers per chunk:
        // three chunks did not fit under the 168-register cap of a nine-wave workgroup)
        ps_u32x4 q[DC][UPW], h[DC];
        __shared__ __attribute__((aligned(16))) char hscr[NW][32 * G4_HX];
        const uint32_t lane16 = (uint32_t)lane * 16u;
        // loads are UNCONDITIONAL (a slot past the range re-reads the workgroup's first unit) so that the compiler counts
        // vmcnt exactly and a chunk is consumed while the next ones are in flight
        auto issue = [&](ps_u32x4 (&q)[UPW], ps_u32x4 &h, int tl, int un, int i_lo = 0, int i_hi = UPW) { // units [i_lo, i_hi) of the slot; the header piece rides with unit 0
            const bool live = tl < nt;
            int grp = t0 + (live ? tl : 0), ul = live ? un : 0;
            const uint8_t *qb = p.w[0].qs, *ab = p.w[0].aux;
            if (EPI == 1) {
                if (ul >= n_units) { ul -= n_units; qb = p.w[1].qs; ab = p.w[1].aux; }
            } else if (p.n_w > 1 && grp >= p.w[0].n_groups) {
                grp -= p.w[0].n_groups; qb = p.w[1].qs; ab = p.w[1].aux;
                if (p.n_w > 2 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; qb = p.w[2].qs; ab = p.w[2].aux; }
            }
            const uint32_t idx = (uint32_t)(grp * n_units + ul);
            const uint8_t *qg = qb + ((uint64_t)idx << 10), *ag = ab + ((uint64_t)idx << 7);
            // a dead slot costs one cache line: every lane asks for the same 16 bytes of the workgroup's first unit
            const uint32_t lo = live ? lane16 : 0u, st = live ? 1u : 0u;
#pragma unroll
            for (int i = 0; i < UPW; i++)
                if (i >= i_lo && i < i_hi) q[i] = __builtin_nontemporal_load((const ps_u32x4 *)(qg + i * (1024 * st) + lo));
            if (i_lo == 0) h = *(const ps_u32x4 *)(ag + (live ? (uint32_t)(lane & 31) * 16u : 0u)); // (plain: with the non-temporal hint the fused launch gains 0.3 us, these launches nothing -- ps_gemv_dev.h G4_HDR_NT)
        };
        constexpr int YN = YS ? YS : 1;
        int4 Y0[YN][UPW], Y1[YN][UPW];
        int2 YB[YN][UPW];
        float YD[YN][UPW];
        int yset = 0; // (a constant once the ring loop is unrolled)
        auto produce = [&](const ps_u32x4 (&q)[UPW], const ps_u32x4 &hc, int tl, int un, int buf, int i_lo = 0, int i_hi = UPW) {
            if (tl >= nt) return; // wave-uniform: nothing of this chunk belongs to the wave
            const int ul = (EPI == 1 && un >= n_units) ? un - n_units : un;
            if (i_lo == 0 && lane < 32) g4_expand_header(hc, hscr[wave] + lane * G4_HX); // lane l loaded the header of (unit l >> 3, row l & 7)
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                if (i < i_lo || i >= i_hi) continue;
                recs[(buf * UPB + wave * UPW + i) * 64 + lane] = YS ? g4_unit_y(q[i], hscr[wave] + (i * 8 + r) * G4_HX, u, Y0[yset][i], Y1[yset][i], YB[yset][i], YD[yset][i])
                                                                    : g4_unit(q[i], hscr[wave] + (i * 8 + r) * G4_HX, ul + i, u, A); // (lanes u >= 4: .w is not a product, their acc_m is never read)
                if (G4_PAIR == 0 || (i & 1)) __builtin_amdgcn_sched_barrier(0); // one unit (G4_PAIR: two) at a time: interleaving four of them costs registers, hides nothing
            }
        };
        auto advance = [&](int &tl, int &un) {
            tl += step_t; un += step_u;
            if (un >= tot) { un -= tot; tl++; }
        };
        // XW (staged issue): only chunk 0 goes out before the prologue.  XW 1: chunk 1 at the sum-of-squares barrier, XW 2 (production):
        // chunk 1 behind the quantizer as well — at the barrier the CU's queue still holds every wave's chunk 0, the waves
        // stalled ~0.9 us IN that issue, on the critical path of the prologue (activation in LDS after 3.1 instead of 4.0 us;
        // gate/up 17.3 -> 16.1 us).  XW 3 adds the split issue of the steady state (no gain, kept for the sweep).  The CU's vector-memory queue is served in order at
        // 64 B per clock and a wave whose requests do not fit stalls IN the issue: with two chunks (16 KiB + headers per
        // wave) up front the waves spent 1.2 us issuing before they could look at the activation row, and the slowest
        // reached the sum-of-squares barrier 1 us after the first.  The later chunks follow behind the prologue's barriers.
        constexpr bool STAGED = XW && PRO != 0;
        issue(q[0], h[0], tS[0], uS[0]);
        if (!STAGED) {
#pragma unroll
            for (int d = 1; d < DC; d++) issue(q[d], h[d], tS[d], uS[d]);
        }
        mark(); // 1: loads issued
        // 3. activation -> LDS (the chain wave joins the barriers)
        if (PRO == 0) {
            for (int i = threadIdx.x; i < K / 4; i += NW * 64) { // (quad-major tiles, as the quantizer writes them)
                const int dw = i & 63;
                ((int *)lq)[(i & ~63) + (((dw & 7) << 3) | (dw >> 3))] = ((const int *)p.aq)[i];
            }
            for (int i = threadIdx.x; i < n_units; i += NW * 64) ld[i] = p.ad[i];
            for (int i = threadIdx.x; i < nb32; i += NW * 64) lb[i] = (int)p.abs16[2 * i] + (int)p.abs16[2 * i + 1];
            __syncthreads();
        } else {
            // RMSNorm (PRO 1: ggml.c:12667-12720, double sum of squares, scale = 1/sqrtf(mean + eps), y = x * (w * scale)) and
            // Q8_K quantization of this wave's tiles (tile t = wave + i * NW); timeline events 24..27 of wave 0 land in the
            // chain role's slots 24..27
            auto pmark = [&](int k) { if (dbg && wave == 0) dbg[32 + k] = __builtin_amdgcn_s_memtime(); };
            float scale = 1.0f;
            if (PRO == 1) {
                double ss = 0.0;
#pragma unroll
                for (int i = 0; i < PPW; i++) {
                    if (wave + i * NW < n_pairs) { // (wave-uniform)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            ss += (double)__fmul_rn(xv[i][h].x, xv[i][h].x);
                            ss += (double)__fmul_rn(xv[i][h].y, xv[i][h].y);
                            ss += (double)__fmul_rn(xv[i][h].z, xv[i][h].z);
                            ss += (double)__fmul_rn(xv[i][h].w, xv[i][h].w);
                        }
                    }
                }
                pmark(24); // the row has arrived
                ss = wave_sum_d_dpp(ss);
                if (lane == 0) red[wave] = ss;
                __syncthreads();
                pmark(25);
                if (STAGED && XW == 1 && DC > 1) issue(q[1], h[1], tS[1], uS[1]); // (XW 2, 3: after the quantizer instead, with the others)
                double tot_ss = 0.0;
#pragma unroll
                for (int i = 0; i <= NW; i++) tot_ss += red[i];
                const float mean = (float)(tot_ss / (double)K);
                scale            = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, p.eps)));
                pmark(26);
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
                g4_quantize_pair(v, tp * 512 + lane * 8, tp, lq, ld, lb, live);
            }
            pmark(27);
            if (STAGED) {
#pragma unroll
                for (int d = ((PRO == 1 && XW == 1) ? 2 : 1); d < DC; d++) issue(q[d], h[d], tS[d], uS[d]);
            }
            __syncthreads();
        }
        mark(); // 2: activation in LDS
        if (YS) { // this wave's activation operands: set k = the units of chunk k
#pragma unroll
            for (int k = 0; k < YN; k++) {
                const int un = uS[k], ul = (EPI == 1 && un >= n_units) ? un - n_units : un;
#pragma unroll
                for (int i = 0; i < UPW; i++) {
                    Y0[k][i] = *(const int4 *)(A.q32 + (ul + i) * 64 + u * 8);
                    Y1[k][i] = *(const int4 *)(A.q32 + (ul + i) * 64 + u * 8 + 4);
                    YB[k][i] = *(const int2 *)(A.bs32 + (ul + i) * 8 + 2 * (u & 3));
                    YD[k][i] = A.d[ul + i];
                }
            }
        }
        // 4. chunk it * DC + d from ring slot d
        for (int it = 0; it < n_iters; it++) {
#pragma unroll
            for (int d = 0; d < DC; d++) {
                yset = YS == 2 ? d : 0;
                if constexpr (XW == 3) {
                    // split issue: the registers of a slot's first two units are free once those units are produced, so
                    // their next loads go out half a chunk earlier (more bytes in flight while this chunk is being produced,
                    // without a deeper ring)
                    const int tl_now = tS[d], un_now = uS[d];
                    advance(tS[d], uS[d]);
                    produce(q[d], h[d], tl_now, un_now, (it * DC + d) & 1, 0, 2);
                    issue(q[d], h[d], tS[d], uS[d], 0, 2);
                    produce(q[d], h[d], tl_now, un_now, (it * DC + d) & 1, 2, 4);
                    mark(); // 3, 6, ...: chunk produced
                    issue(q[d], h[d], tS[d], uS[d], 2, 4);
                } else {
                    produce(q[d], h[d], tS[d], uS[d], (it * DC + d) & 1);
                    mark(); // 3, 6, ...: chunk produced
                    advance(tS[d], uS[d]);
                    issue(q[d], h[d], tS[d], uS[d]);
                }
                mark(); // 4, 7, ...: next loads issued
                __syncthreads();
                mark(); // 5, 8, ...: barrier passed
            }
        }
    } else { // ------------------------------------------------------------------------- chain wave
        // Everything the epilogue reads from memory is fetched NOW, while this wave has nothing to do, into LDS: a global
        // load on the chain's path costs a vector-memory round trip behind the weight stream (> 1 us) per row group
        // (measured: 2 us per chunk of the gate/up launch with the expf table in constant memory, 7 chunks per launch).
        const int ep_n = (p.split_q + 1) * 8;
        float *const epB = epA + ep_n, *const epC = epB + ep_n;
        int kv_pos = 0, rpos = 0;
        if (EPI == 2) { kv_pos = p.rope.state->pos0; rpos = p.rope.rope_pos ? p.rope.rope_pos[0] : kv_pos; }
        if (PRO == 1) { // the sum-of-squares exchange first: nobody waits for this wave's loads there
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
                if (p.n_w > 1 && grp >= p.w[0].n_groups) { grp -= p.w[0].n_groups; wi = 1; }
                if (p.n_w > 2 && wi == 1 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; wi = 2; }
                const int64_t Nw = wi == 0 ? p.w[0].N : (wi == 1 ? p.w[1].N : p.w[2].N);
                const float *b   = wi == 0 ? p.w[0].bias : (wi == 1 ? p.w[1].bias : p.w[2].bias);
                const int64_t row = (int64_t)grp * TR::RG + r;
                float va = 0.f, vb = 0.f, vc = 0.f;
                if (row < Nw) {
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
                }
                epA[tl * 8 + r] = va; epB[tl * 8 + r] = vb; epC[tl * 8 + r] = vc;
            }
        }
        __syncthreads();
        mark(); // 1: activation in LDS
        __builtin_amdgcn_s_setprio(3); // one wave serves NW producers: it gets the issue slots first
        float acc0 = 0.f, acc1 = 0.f, accm = 0.f, ygate = 0.f;
        int tl = 0, un = 0; // local task, units of it already chained
        auto row_done = [&]() {
            const float y = row_reduce<WT>(acc0, acc1, accm);
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
            const int64_t row = (int64_t)grp * TR::RG + r;
            const float ea = EPI != 1 ? epA[tl * 8 + r] : 0.f, eb = EPI == 2 ? epB[tl * 8 + r] : 0.f, ec = EPI != 1 ? epC[tl * 8 + r] : 0.f;
            if constexpr (EPI == 2) { // q / k: rotate adjacent pairs (rows 2i, 2i+1 sit in neighbouring lane groups); v: transpose-append
                float v = y;
                if (b && row < Nw) v = __fadd_rn(v, ec);
                const float vp = dpp_f<0x128>(v); // partner row (row_ror:8 swaps the two row groups of 8 lanes)
                const psk_rope_kv &R = p.rope;
                const int role = wi + p.rope_wi0; // 0 q, 1 k, 2 v
                if (u == 0 && row < Nw) {
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
            } else if (u == 0 && row < Nw) {
                if (EPI == 1) {
                    g4_out(o + row, g4_silu_mul(ygate, y, exp_tab));
                } else {
                    float v = y;
                    if (b) v = __fadd_rn(v, ec);
                    if (p.residual && wi == 0) v = __fadd_rn(ea, v);
                    g4_out(o + row, v);
                }
            }
            acc0 = 0.f; acc1 = 0.f; accm = 0.f;
            un = 0;
            tl++;
        };
        auto batch = [&](auto nconst, const Rec *rb, const int k0) { // N records in one LDS round trip, then the two fma chains
            constexpr int N = decltype(nconst)::value;
            Rec rc[N];
#pragma unroll
            for (int k = 0; k < N; k++) rc[k] = rb[(k0 + k) * 64];
#pragma unroll
            for (int k = 0; k < N; k++) {
                acc0 = __fmaf_rn(rc[k].x, rc[k].y, acc0);
                accm = __fmaf_rn(rc[k].z, rc[k].w, accm); // lanes u >= 4: not an acc_m lane, never read
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int c = 0; c < DC * n_iters; c++) {
            mark(); // 2, 4, ...: previous chunk chained, waiting
            __syncthreads();
            mark(); // 3, 5, ...: chunk c handed over
            if (c >= n_chunks) continue;
            const Rec *rb  = recs + (size_t)(c & 1) * UPB * 64 + lane;
            const int kend = min(UPB, s_end - c * UPB);
            for (int k0 = 0; k0 < kend;) { // runs: units of one row (EPI 1: of one half of a gate/up pair); lengths are multiples of 4
                const int bound = (EPI == 1 && un < n_units) ? n_units : tot;
                const int len   = min(bound - un, kend - k0);
                int kk = k0, rem = len;
                for (; rem >= 16; rem -= 16, kk += 16) batch(std::integral_constant<int, 16>{}, rb, kk);
                if (rem >= 8) { batch(std::integral_constant<int, 8>{}, rb, kk); rem -= 8; kk += 8; }
                if (rem >= 4) batch(std::integral_constant<int, 4>{}, rb, kk);
                un += len;
                k0 += len;
                if (EPI == 1 && un == n_units) { // gate row finished: reduce it, restart the chains for the up row
                    ygate = row_reduce<WT>(acc0, acc1, accm);
                    acc0 = 0.f; acc1 = 0.f; accm = 0.f;
                }
                if (un == tot) row_done();
            }
        }
    }
    if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

template <int NW, int DC, int TPW, int XW, int EPI, int PRO, int YS = 0>
void launch_g4(hipStream_t st, int grid, const G4Params &p) {
    const size_t smem = (size_t)p.col_bytes + (size_t)2 * NW * 4 * 64 * sizeof(float4) + (size_t)3 * (p.split_q + 1) * 8 * sizeof(float);
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr) && smem > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)gemv4_kernel<NW, DC, TPW, XW, EPI, PRO, YS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    }
    if (YS) psk_note_kernel("gemv4_kernel<%d, %d, %d, %d, %d, %d, %d>", NW, DC, TPW, XW, EPI, PRO, YS);
    else psk_note_kernel("gemv4_kernel<%d, %d, %d, %d, %d, %d>", NW, DC, TPW, XW, EPI, PRO);
    hipLaunchKernelGGL((gemv4_kernel<NW, DC, TPW, XW, EPI, PRO, YS>), dim3((unsigned)grid), dim3((NW + 1) * 64), smem, st, p);
}

// KC: row-length class (tiles of 256 per row <= 16 << KC)
template <int NW, int DC, int XW, int KC, int YS = 0>
int launch_g4_ep(hipStream_t st, int grid, const G4Params &p, int epi, int pro) {
    constexpr int TPW = ((16 << KC) + NW - 1) / NW;
    if (epi == 2) { if (pro != 1) return -1; launch_g4<NW, DC, TPW, XW, 2, 1, YS>(st, grid, p); return 0; }
    if (epi == 1) {
        if (pro == 1) launch_g4<NW, DC, TPW, XW, 1, 1, YS>(st, grid, p);
        else if (pro == 0) launch_g4<NW, DC, TPW, XW, 1, 0, YS>(st, grid, p);
        else return -1;
        return 0;
    }
    if (pro == 0) launch_g4<NW, DC, TPW, XW, 0, 0, YS>(st, grid, p);
    else if (pro == 1) launch_g4<NW, DC, TPW, XW, 0, 1, YS>(st, grid, p);
    else launch_g4<NW, DC, TPW, XW, 0, 2, YS>(st, grid, p);
    return 0;
}
template <int NW, int DC, int XW, int YS = 0>
int launch_g4_kc(hipStream_t st, int grid, const G4Params &p, int epi, int pro) {
    if (p.n_units <= 16) return launch_g4_ep<NW, DC, XW, 0, YS>(st, grid, p, epi, pro);
    if (p.n_units <= 64) return launch_g4_ep<NW, DC, XW, 2, YS>(st, grid, p, epi, pro);
    return -1;
}

} // namespace

int g_g4_cfg = getenv("PS_G4_CFG") ? atoi(getenv("PS_G4_CFG")) : 0; // ps_hip_debug_set(1, cfg)  (cfg 20..23 were round 4's LDS-DMA kernel: tools/experiments/r04_k_gemv7.hip)
int g_g4_flags = 0; // ps_hip_debug_set(2, flags): reserved for what-if switches

bool psk_gemv4_covers(int64_t K) { // rows end on multiples of four units
    static const bool off = getenv("PS_NO_GEMV4") != nullptr; // (A/B switch for measurements)
    return !off && K % 1024 == 0 && K <= 16384;
}

// Single-column Q4_K mat-vec.  Returns -1 when the launch is not covered (the caller falls back to gemv1 / gemv_kernel).
int psk_gemv4(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K) {
    if (a.n_w < 1 || a.n_w > 3 || !psk_gemv4_covers(K)) return -1;
    G4Params p{};
    int groups_total = 0;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != PS_Q4_K || a.w[i]->K != K) return -1;
        const int ng = (int)((a.w[i]->N + 7) / 8);
        p.w[i] = G4Mat{a.w[i]->qs, a.w[i]->aux, a.out[i], a.bias[i], a.w[i]->N, ng};
        groups_total += ng;
    }
    const int epi = a.silu_pair ? 1 : (a.rope ? 2 : 0);
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N)) return -1;
    if (epi == 2) {
        if (a.n_w + a.rope_wi0 > 3 || a.rope_wi0 < 0 || a.pro != 1) return -1;
        p.rope = *a.rope; p.rope_wi0 = a.rope_wi0;
    }
    p.n_w = a.n_w; p.n_units = (int)(K / 256); p.K = (int)K;
    p.n_tasks = epi == 1 ? p.w[0].n_groups : groups_total;
    p.residual = a.residual; p.x = a.pro_x; p.nw = a.pro_norm_w; p.eps = a.pro_eps;
    p.aq = act.qs; p.ad = act.d; p.abs16 = act.bs16;
    p.col_bytes = (int)psk_gemv_lds_col_bytes(PS_Q4_K, K);
    int grid = p.n_tasks < n_cu ? p.n_tasks : n_cu;
    if (grid < 1) return -1;
    p.split_q = p.n_tasks / grid; p.split_r = p.n_tasks % grid;
    p.dbg = psk_gemv_dbg_buf(epi, (epi == 0 && a.pro == 2 && K <= a.w[0]->N) ? 3 : a.pro); // timeline keys: 9 QKV, 3 O (K <= N), 5 gate/up, 2 down
    // wave configuration (measured, tools/g4_variants.py): rows of a multiple of 7 units take 7 (14) producers
    const bool seven = p.n_units % 7 == 0;
    if (g_g4_cfg == 40 || g_g4_cfg == 41) { // register-resident activation operands on the register ring (41: three chunks in flight)
        const int tot = (epi == 1 ? 2 : 1) * p.n_units, upb = (seven ? 7 : 8) * 4;
        const int ys = upb % tot == 0 ? 1 : ((2 * upb) % tot == 0 ? 2 : 0);
        if (ys == 1 && g_g4_cfg == 41 && !seven) return launch_g4_kc<8, 3, 2, 1>(st, grid, p, epi, a.pro);
        if (ys == 1) return seven ? launch_g4_kc<7, 2, 2, 1>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 2, 2, 1>(st, grid, p, epi, a.pro);
        if (ys == 2) return seven ? launch_g4_kc<7, 2, 2, 2>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 2, 2, 2>(st, grid, p, epi, a.pro);
    }
    switch (g_g4_cfg) { // (0 is the production configuration; the others are kept for tools/g4_variants.py)
    case 1: return seven ? launch_g4_kc<7, 2, 0>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 2, 0>(st, grid, p, epi, a.pro); // everything issued up front
    case 2: return seven ? launch_g4_kc<7, 3, 1>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 3, 1>(st, grid, p, epi, a.pro); // three chunks in flight
    case 3: return launch_g4_kc<11, 2, 1>(st, grid, p, epi, a.pro);                                                          // twelve waves
    case 12: return launch_g4_kc<10, 2, 2>(st, grid, p, epi, a.pro);
    case 13: return launch_g4_kc<11, 2, 2>(st, grid, p, epi, a.pro);
    case 14: return launch_g4_kc<12, 2, 2>(st, grid, p, epi, a.pro);
    case 7: return seven ? launch_g4_kc<7, 2, 3>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 2, 3>(st, grid, p, epi, a.pro); // + split issue
    case 6: return seven ? launch_g4_kc<7, 3, 2>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 3, 2>(st, grid, p, epi, a.pro);
    case 4: return seven ? launch_g4_kc<7, 4, 1>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 4, 1>(st, grid, p, epi, a.pro); // four chunks in flight
    case 8: return seven ? launch_g4_kc<7, 2, 1>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 2, 1>(st, grid, p, epi, a.pro); // chunk 1 at the sum-of-squares barrier (round-2 default until the timeline showed the stall)
    default:
        // register-resident activation operands (YS = 1) where a wave meets the same four super-blocks in every chunk and the stream is long
        // enough to pay for fetching them once: gate/up and the lm_head of a K = 4096 model (round 4, profiles/r04_gemv_variants.txt:
        // gate/up 16.1 -> 15.4 us, lm_head 51.4 -> 50.0 us with 8 producers; the one- and two-chunk launches did not move)
        if (!seven && (8 * 4) % ((epi == 1 ? 2 : 1) * p.n_units) == 0 && (epi == 1 || (a.n_w == 1 && p.split_q >= 32)) && g_g4_cfg != 42)
            return launch_g4_kc<8, 2, 2, 1>(st, grid, p, epi, a.pro);
        // eleven producers where the stream per workgroup is long and there is one matrix (down: K = 14336; lm_head: 63 row
        // groups per workgroup): 45 KB per chunk in flight instead of 29-37 (down 12.2 -> 11.6 us, lm_head 54.5 -> 51 us);
        // the short launches and gate/up measured slower with them (QKV 7.5 -> 9.1 us: prologue and boundary of 768 threads)
        if (epi == 0 && a.n_w == 1 && (seven || p.split_q >= 32)) return launch_g4_kc<11, 2, 2>(st, grid, p, epi, a.pro);
        return seven ? launch_g4_kc<7, 2, 2>(st, grid, p, epi, a.pro) : launch_g4_kc<8, 2, 2>(st, grid, p, epi, a.pro);
    }
}
