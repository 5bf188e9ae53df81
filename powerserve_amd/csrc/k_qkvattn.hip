// One launch for the head of a decode layer (round 6): the Q / K / V mat-vec (RMSNorm + Q8_K quantizer prologue, RoPE + KV-append epilogue: gemv4_kernel<8, 2, 2, 2, 2, 1>
// of k_gemv4.hip) AND the single-token attention (attn_decode2_kernel<NV, 512> of k_attn.hip), four launches per layer instead of five.
// Reference: src/model/module/norm_attention.cpp:62-147 (RMS_NORM, three MAT_MULs, ROPE x 2, the KV append, K.q, SOFTMAX_EXT, V.p, PERMUTE + CONT); the arithmetic
// of both halves is the unfused kernels', statement by statement, so the bits are theirs.
//
// Why this edge and no other: Q / K / V -> attention is the only edge of a layer that is NOT an all-to-all.  A kv head's attention needs that head group's rows only
// (r2 q heads + k + v = (r2 + 2) * head_size rows), the one-launch attention already gives a kv head head_size / 4 workgroups (4 V channels each) that meet at a
// per-head counter, and with 8 kv heads those are the 32 CUs of one XCD.  So workgroup (kv head h, slice b) takes ITS share of head group h's row groups, the kv head's
// workgroups meet once more -- rendezvous A, after the rotated q, the new K row and the new V column have been stored write-through -- and go on to the scores.
// What the fusion buys (profiles/r06_boundaries.txt: a kernel boundary is 1.5-1.7 us, the attention's first 2 us wait for the device-resident position and only then ask
// for K and V): one boundary less, and the cached K rows / V channels -- which do not depend on this step's Q / K / V at all -- are requested behind the quantizer, a whole
// weight stream before they are needed.  (Split-K over O is not available: it would reorder the reference's fp32 chain.)
//
// Workgroup: 9 waves (a kernel boundary behind 832-thread workgroups costs ~1 us more than behind 576-thread ones: the second version, with four cache waves of
// its own, lost there what it had won).  Mat-vec phase: waves 0-7 producers, wave 8 the chain wave (k_gemv4.hip).  The producers ask for the cached K rows (registers,
// attn_decode2's layout) and the four V channels (global -> LDS without registers, global_load_lds_dwordx4) when their last chunk is produced.  Attention phase:
// waves 0-7 are attn_decode2's 512 threads, wave 8 takes the tickets and polls.  LDS: [small attention arrays][V rows 4 x RS][shared: mat-vec image + records + headers | e rows 4 x RS].
//
// Visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): everything one workgroup hands another inside the launch is stored
// write-through (sc1), drained, then ticketed; the reader polls ONE word, and the lines it then reads with plain loads are lines this launch has not touched before
// (the new K row, the 128-byte lines of the V rows that hold the new column, q, the score rows) -- so neither an L1 nor an L2 anywhere holds an older copy.  That is why
// the cached K rows are requested only BELOW the position and the V rows only up to the last whole line below it.  Placement (block b on XCD b % 8) is for speed only.
#include "ps_expf.h"
#include "ps_g4_dev.h"
#include "ps_ops.h"
// KVS (psl_attn_args::kv_stream, round 6): the cached K rows and V channels are read with NON-TEMPORAL loads, and the wave that owns the new position then asks for the new row
// only, not for its seven streamed neighbours again.  They are read once per token; when the model's whole cache is larger than the memory-side cache (8B at n_kv 2048: 537 MB)
// nothing of it survives to the next token and, read with plain loads, it displaces everything else there and in the L2s: same-box A/B, 8B decode over 32 steps
// (profiles/r06_kv_nt_ab.txt): 577.5 tok/s plain, K 584.6, V 582.6, both 587.1, with the new-row request 589.8 (+2.1 %).  A cache that fits stays there: plain loads (k_attn.hip).

namespace {
typedef float ps_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void qa_store_f(float *p, float v) { __hip_atomic_store((uint32_t *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct QAParams {
    const uint8_t *qs[3], *aux[3]; // Wq, Wk, Wv: lane-major quants, headers
    const float *bias[3];
    int n_units, K, col_bytes, rec_units;
    const float *x, *nw;           // RMSNorm(x, nw, eps), then Q8_K
    float eps;
    int nq, nk;                    // row groups of ONE kv head in Wq and in Wk (= Wv)
    int split_q, split_r;          // tasks of a workgroup: split_q (+ 1 for the first split_r workgroups of its kv head)
    psl_attn_args a;
};

constexpr int QA_MAXCTX = 4096, QA_THREADS = 9 * 64;

// One bounded rendezvous of the G workgroups of a kv head on a counter that is never reset (attn_decode2's exchange, epoch = count / G).  `base` is the counter as this
// workgroup read it at kernel ENTRY: every earlier launch is complete (a multiple of G) and of this launch fewer than G workgroups can have arrived before this one
// has, so base / G IS this launch's epoch whatever the timing -- the arrival itself is then an atomic add nobody waits for (a returned ticket is a memory round
// trip on the critical path: 2.0 us from "rows stored" to "rendezvous complete" in profiles/r06_fused_timeline_v1.txt).
__device__ __forceinline__ void qa_rendezvous(unsigned *ctr, const unsigned base, const unsigned G, unsigned *flag) {
    (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (base / G + 1u) * G;
    int spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 16)) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; } // never hang the GPU: the host turns the flag into a re-run on the unfused launches
    }
}

template <int NV, int KVS> // head_size / 32; the cache is streamed (non-temporal loads: see above)
__global__ __launch_bounds__(QA_THREADS) void qkv_attn_kernel(const QAParams p) {
    constexpr int NW = 8, DC = 2, UPW = 4, UPB = NW * UPW;                                       // mat-vec geometry (gemv4_kernel<8, 2, ...>)
    constexpr int NT = 512, AW = NT / 64, SPP = AW / 4, LPH = NT / 4, WPH = LPH / 64, D2_TRIPS = QA_MAXCTX / (8 * LPH); // attention geometry (attn_decode2_kernel<NV, 512>)
    constexpr int hs = NV * 32, RMAX = 16 / NV, G = hs / 4, PASSES = RMAX / SPP; // RMAX: 32-position slices of a workgroup; a pass = 8 waves x 8 positions = 2 slices
    using Rec = float4;
    const psl_attn_args &a = p.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kvd = a.n_kv_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int kvh = (int)blockIdx.x % a.n_kv_heads, bx = (int)blockIdx.x / a.n_kv_heads;
    const int RS = ((a.n_ctx + 127) & ~127) + 36, XS = (a.n_ctx + 31) & ~31;
    const int K = p.K, n_units = p.n_units, nb32 = K / 32;
    // ---- LDS
    float *const red = (float *)smem, *const pleft = red + 512, *const redf = pleft + 128, *const tails = redf + 16, *const qs = tails + 32, *const sst = qs + 4 * hs;
    double *const redd = (double *)(sst + RMAX * 128), *const red_ss = redd + 16;
    uint64_t *const etab = (uint64_t *)(red_ss + 16);
    float *const vt = (float *)(etab + PS_EXP2F_N);      // [4][RS] this workgroup's V channels
    char *const big = (char *)(vt + 4 * RS);
    float *const pl = (float *)big;                       // attention phase: [4][RS] e_j
    int8_t *const lq = (int8_t *)big;                     // mat-vec phase: the activation column's Q8_K image ...
    float *const ld = (float *)(big + K);
    int *const lb = (int *)(ld + n_units);
    Rec *const recs = (Rec *)(big + p.col_bytes);         // ... the records of the workgroup's whole stream [rec_units][64] ...
    char *const hscr = (char *)(recs + (size_t)p.rec_units * 64); // ... a slot's expanded headers per producer [NW][32 * G4_HX] ...
    float *const epA = (float *)(hscr + NW * 32 * G4_HX); // ... and the epilogue's operands [3][(split_q + 1) * 8]
    LAct A;
    A.q32 = (const int *)lq; A.d = ld; A.bs32 = lb;
    unsigned long long *const dbg = (PS_TL(a.dbg) && blockIdx.x < 1024 && (tid == 0 || tid == NT)) ? a.dbg + (size_t)blockIdx.x * 64 + (tid == NT ? 32 : 0) : nullptr; // timeline key 43
    auto mark = [&](int k) { if (dbg) dbg[k] = __builtin_amdgcn_s_memtime(); };
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }

    // ---- t = 0: the position (a vector load on purpose, cf. attn_decode2_kernel), first of everything: it lands with the activation row
    const ps_step_state *sp = a.state;
    asm volatile("" : "+v"(sp));
    const int st_pos0 = *(const __attribute__((address_space(1))) int *)(uintptr_t)sp;

    // this workgroup's tasks: local row groups [t0, t0 + nt) of its kv head's list {Wq groups, Wk groups, Wv groups}
    const int t0 = bx * p.split_q + min(bx, p.split_r), nt = p.split_q + (bx < p.split_r ? 1 : 0);
    const int tot = n_units, s_end = nt * tot, n_chunks = (s_end + UPB - 1) / UPB, n_iters = (n_chunks + DC - 1) / DC;
    auto task_of = [&](int li, int &wi, int &grp) { // local task -> (matrix, row group of that matrix)
        if (li < p.nq) { wi = 0; grp = kvh * p.nq + li; }
        else if (li < p.nq + p.nk) { wi = 1; grp = kvh * p.nk + (li - p.nq); }
        else { wi = 2; grp = kvh * p.nk + (li - p.nq - p.nk); }
    };
    const int r = lane >> 3, u = lane & 7;
    const int p8 = lane >> 3, tq = lane & 7; // attention: eight lanes per cached position; lane tq owns chains 4 tq .. 4 tq + 3
    const float *kb = a.k_cache + kvh * hs + 4 * tq;
    const float *vbase = a.v_cache + ((int64_t)kvh * hs + bx * 4) * a.n_ctx;
    int pos0 = 0;
    float4 kf[PASSES][NV]; // waves 0-7: K row of this lane's position in each pass
    unsigned baseA = 0, baseB = 0; // the rendezvous counters as they stood at entry (wave 8, lane 0: qa_rendezvous)

    if (wave < NW) { // ================================================================== mat-vec phase: producers (k_gemv4.hip, PRO 1, XW 2, YS 0)
        constexpr int PPW = 1; // K <= 4096: 8 pairs of tiles, one per producer
        const int n_pairs = n_units / 2;
        float4 xv[PPW][2], wv[PPW][2];
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            const int tp = wave + i * NW;
            const int64_t e = (int64_t)(tp < n_pairs ? tp : 0) * 512 + lane * 8;
            xv[i][0] = *(const float4 *)(p.x + e); xv[i][1] = *(const float4 *)(p.x + e + 4);
            wv[i][0] = *(const float4 *)(p.nw + e); wv[i][1] = *(const float4 *)(p.nw + e + 4);
        }
        const int step_t = (DC * UPB) / tot, step_u = (DC * UPB) % tot;
        int tS[DC], uS[DC];
#pragma unroll
        for (int d = 0; d < DC; d++) {
            int t = 0, un = d * UPB + wave * UPW;
            while (un >= tot) { un -= tot; t++; }
            tS[d] = t; uS[d] = un;
        }
        ps_u32x4 q[DC][UPW], h[DC];
        const uint32_t lane16 = (uint32_t)lane * 16u;
        auto issue = [&](ps_u32x4 (&q)[UPW], ps_u32x4 &h, int tl, int un) { // UNCONDITIONAL loads (a dead slot re-reads the workgroup's first unit): exact vmcnt counts
            const bool live = tl < nt;
            int wi, grp;
            task_of(t0 + (live ? tl : 0), wi, grp);
            const int ul = live ? un : 0;
            const uint8_t *qb = wi == 0 ? p.qs[0] : (wi == 1 ? p.qs[1] : p.qs[2]), *ab = wi == 0 ? p.aux[0] : (wi == 1 ? p.aux[1] : p.aux[2]);
            const uint32_t idx = (uint32_t)(grp * n_units + ul);
            const uint8_t *qg = qb + ((uint64_t)idx << 10), *ag = ab + ((uint64_t)idx << 7);
            const uint32_t lo = live ? lane16 : 0u, st = live ? 1u : 0u;
#pragma unroll
            for (int i = 0; i < UPW; i++) q[i] = __builtin_nontemporal_load((const ps_u32x4 *)(qg + i * (1024 * st) + lo));
            h = G4_HDR_NT ? __builtin_nontemporal_load((const ps_u32x4 *)(ag + (live ? (uint32_t)(lane & 31) * 16u : 0u))) : *(const ps_u32x4 *)(ag + (live ? (uint32_t)(lane & 31) * 16u : 0u));
        };
        auto produce = [&](const ps_u32x4 (&q)[UPW], const ps_u32x4 &hc, int tl, int un, int chunk) {
            if (tl >= nt) return; // wave-uniform
            char *hx = hscr + wave * (32 * G4_HX);
            if (lane < 32) g4_expand_header(hc, hx + lane * G4_HX);
#pragma unroll
            for (int i = 0; i < UPW; i++) {
                recs[(size_t)(chunk * UPB + wave * UPW + i) * 64 + lane] = g4_unit(q[i], hx + (i * 8 + r) * G4_HX, un + i, u, A);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto advance = [&](int &tl, int &un) {
            tl += step_t; un += step_u;
            if (un >= tot) { un -= tot; tl++; }
        };
        issue(q[0], h[0], tS[0], uS[0]);
        mark(1); // loads issued
        // RMSNorm (ggml.c:12667-12720) + Q8_K (ggml-quants.c:3799-3835) of this wave's pair of tiles
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            if (wave + i * NW < n_pairs) {
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    ss += (double)__fmul_rn(xv[i][hh].x, xv[i][hh].x);
                    ss += (double)__fmul_rn(xv[i][hh].y, xv[i][hh].y);
                    ss += (double)__fmul_rn(xv[i][hh].z, xv[i][hh].z);
                    ss += (double)__fmul_rn(xv[i][hh].w, xv[i][hh].w);
                }
            }
        }
        ss = wave_sum_d_dpp(ss);
        if (lane == 0) red_ss[wave] = ss;
        __syncthreads(); // #1
        double tot_ss = 0.0;
#pragma unroll
        for (int i = 0; i <= NW; i++) tot_ss += red_ss[i];
        const float mean  = (float)(tot_ss / (double)K);
        const float scale = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, p.eps)));
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            const int tp = wave + i * NW;
            const bool live = tp < n_pairs;
            float v[8] = {xv[i][0].x, xv[i][0].y, xv[i][0].z, xv[i][0].w, xv[i][1].x, xv[i][1].y, xv[i][1].z, xv[i][1].w};
            const float w8[8] = {wv[i][0].x, wv[i][0].y, wv[i][0].z, wv[i][0].w, wv[i][1].x, wv[i][1].y, wv[i][1].z, wv[i][1].w};
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = __fmul_rn(v[k], __fmul_rn(w8[k], scale));
            g4_quantize_pair(v, tp * 512 + lane * 8, tp, lq, ld, lb, live);
        }
        mark(2); // activation quantized
#pragma unroll
        for (int d = 1; d < DC; d++) issue(q[d], h[d], tS[d], uS[d]);
        pos0 = __builtin_amdgcn_readfirstlane(st_pos0);
        mark(3); // second chunk requested
        __syncthreads(); // #2
        mark(4);
        // The cached K rows below the position (registers: eight lanes per position, attn_decode2's layout) do not depend on this step: a producer asks for its
        // share when ITS last live chunk is produced (K = 4096: waves 4-7 a chunk before waves 0-3) -- behind every weight request, in front of nothing it waits for.
        // (Requested behind the second weight chunk they cost the prologue 1.5 us IN the issue -- the CU's memory queue is served in order and 64 KB per CU take the
        // chip 2.8 us to stream -- and the weight stream its bandwidth: profiles/r06_fused_timeline_v1.txt; all at the end they held up the chain wave's drain: ..._v3.txt.)
        auto request_k = [&]() {
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++) {
                const int sl = bx + ((wave >> 2) + SPP * ps) * G, j0 = sl * 32 + (wave & 3) * 8, j = j0 + p8;
                if (j0 < pos0) { // (uniform) at least one cached row; lanes at or past the position read row 0 again (an old row: pos0 > 0 here)
                    const float *kr = kb + (int64_t)(j < pos0 ? j : 0) * kvd;
#pragma unroll
                    for (int m = 0; m < NV; m++) {
                        if (KVS) { const ps_u32x4 t = __builtin_nontemporal_load((const ps_u32x4 *)(kr + m * 32)); __builtin_memcpy(&kf[ps][m], &t, 16); }
                        else kf[ps][m] = *(const float4 *)(kr + m * 32);
                    }
                }
            }
        };
        const int my_last = s_end > wave * UPW ? (s_end - wave * UPW + UPB - 1) / UPB - 1 : -1; // the last chunk this wave has a live slot in
        if (my_last < 0) request_k();
        for (int it = 0; it < n_iters; it++) {
#pragma unroll
            for (int d = 0; d < DC; d++) {
                produce(q[d], h[d], tS[d], uS[d], it * DC + d);
                if (it * DC + d == my_last) request_k(); // (uniform)
                advance(tS[d], uS[d]);
                issue(q[d], h[d], tS[d], uS[d]);
                __syncthreads(); // chunk it * DC + d handed to the chain wave
            }
        }
        mark(5); // last chunk produced
        __syncthreads(); // A0: the chain wave's rows are stored and drained
        // this workgroup's four V channels up to the last whole 128-byte line below the position: global -> LDS without registers (256 columns of one row per
        // wave-instruction).  HERE: behind the chain wave's drain (the CU's memory pipe is first in, first out across its waves: requested earlier they sat in front of
        // it), while the kv head's workgroups meet.  Not counted by the compiler: drained with the score stores, long before V.p reads them.
        {
            const int vlim = pos0 & ~31; // columns [0, vlim) are lines nobody writes in this launch
            const unsigned vt0 = g4_lds_addr(vt);
            for (int pi = wave; (pi >> 2) * 256 < vlim; pi += NW) { // piece pi: row pi & 3, columns (pi >> 2) * 256 ..
                const int row = pi & 3, c0 = (pi >> 2) * 256, col = c0 + 4 * lane;
                if (col < vlim) {
                    if (KVS) g4_pull_nt((const uint8_t *)(vbase + (int64_t)row * a.n_ctx + col), vt0 + (unsigned)(row * RS + c0) * 4u);
                    else g4_pull((const uint8_t *)(vbase + (int64_t)row * a.n_ctx + col), vt0 + (unsigned)(row * RS + c0) * 4u);
                }
            }
        }
        mark(14); // V requested
    } else if (wave == NW) { // ========================================================= mat-vec phase: the chain wave (EPI 2: RoPE + KV append)
        if (lane == 0) { // (sc1 loads: the counters live in other workgroups' atomics)
            baseA = __hip_atomic_load(a.tick + kvh * 64 + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            baseB = __hip_atomic_load(a.tick + kvh * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int ep_n = (p.split_q + 1) * 8;
        float *const epB = epA + ep_n, *const epC = epB + ep_n;
        pos0 = __builtin_amdgcn_readfirstlane(st_pos0);
        const int kv_pos = pos0, rpos = pos0;
        if (lane == 0) red_ss[wave] = 0.0;
        __syncthreads(); // #1
        for (int tl0 = 0; tl0 < nt; tl0 += 8) { // lane (r, u): row r of local task tl0 + u -- bias, (cos, sin) of the row's rotation pair
            const int tl = tl0 + u;
            if (tl >= nt) continue;
            int wi, grp;
            task_of(t0 + tl, wi, grp);
            const float *b = wi == 0 ? p.bias[0] : (wi == 1 ? p.bias[1] : p.bias[2]);
            const int64_t row = (int64_t)grp * 8 + r;
            float va = 0.f, vb = 0.f, vc = 0.f;
            if (b) vc = b[row];
            if (wi != 2) {
                const int e = (int)(row % hs);
                if (e < a.n_dims) {
                    const int64_t i0 = (int64_t)rpos * hs + (e & ~1);
                    va = a.rope_table[i0]; vb = a.rope_table[i0 + 1];
                }
            }
            epA[tl * 8 + r] = va; epB[tl * 8 + r] = vb; epC[tl * 8 + r] = vc;
        }
        __syncthreads(); // #2
        __builtin_amdgcn_s_setprio(3);
        float acc0 = 0.f, acc1 = 0.f, accm = 0.f;
        int tl = 0, un = 0;
        auto row_done = [&]() {
            const float y = row_reduce<PS_Q4_K>(acc0, acc1, accm);
            int wi, grp;
            task_of(t0 + tl, wi, grp);
            const float *b = wi == 0 ? p.bias[0] : (wi == 1 ? p.bias[1] : p.bias[2]);
            const int64_t row = (int64_t)grp * 8 + r;
            const float ea = epA[tl * 8 + r], eb = epB[tl * 8 + r], ec = epC[tl * 8 + r];
            float v = y;
            if (b) v = __fadd_rn(v, ec);
            const float vp = dpp_f<0x128>(v); // partner row of the rotation pair
            if (u == 0) { // everything another workgroup reads after rendezvous A is stored write-through
                if (wi == 2) {
                    qa_store_f(a.v_cache + row * a.n_ctx + kv_pos, v);
                } else {
                    const int e = (int)(row % hs);
                    float res = v;
                    if (e < a.n_dims) {
                        const float x0 = (e & 1) ? vp : v, x1 = (e & 1) ? v : vp;
                        res = ps_rope_one(x0, x1, ea, eb, (e & 1) != 0);
                    }
                    if (wi == 0) qa_store_f(a.q + row, res); else qa_store_f(a.k_cache + (int64_t)kv_pos * kvd + row, res);
                }
            }
            acc0 = 0.f; acc1 = 0.f; accm = 0.f;
            un = 0;
            tl++;
        };
        auto batch = [&](auto nconst, const Rec *rb, const int k0) {
            constexpr int N = decltype(nconst)::value;
            Rec rc[N];
#pragma unroll
            for (int k = 0; k < N; k++) rc[k] = rb[(k0 + k) * 64];
#pragma unroll
            for (int k = 0; k < N; k++) {
                acc0 = __fmaf_rn(rc[k].x, rc[k].y, acc0);
                accm = __fmaf_rn(rc[k].z, rc[k].w, accm);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int c = 0; c < DC * n_iters; c++) {
            __syncthreads(); // chunk c handed over
            if (c >= n_chunks) continue;
            const Rec *rb  = recs + (size_t)c * UPB * 64 + lane;
            const int kend = min(UPB, s_end - c * UPB);
            for (int k0 = 0; k0 < kend;) {
                const int len = min(tot - un, kend - k0);
                int kk = k0, rem = len;
                for (; rem >= 16; rem -= 16, kk += 16) batch(std::integral_constant<int, 16>{}, rb, kk);
                if (rem >= 8) { batch(std::integral_constant<int, 8>{}, rb, kk); rem -= 8; kk += 8; }
                if (rem >= 4) batch(std::integral_constant<int, 4>{}, rb, kk);
                un += len;
                k0 += len;
                if (un == tot) row_done();
            }
        }
        __builtin_amdgcn_s_setprio(0);
        mark(1); // rows stored
        // ---- rendezvous A: this workgroup's q / k / v are drained, then its ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // A0
        if (lane == 0) qa_rendezvous(a.tick + kvh * 64 + 32, baseA, (unsigned)G, a.sync + 31);
        asm volatile("" ::: "memory");
        mark(2); // rendezvous A complete
    }
    __syncthreads(); // ==== A: q, the new K row and the new V column of this kv head are in memory
    mark(6);

    // ================================================================================== attention phase (attn_decode2_kernel<NV, 512>; wave 8 only keeps the barriers company)
    const bool aw = wave < AW;
    const int uw = wave;
    const int n_kv = pos0 + 1, n8 = n_kv & ~7, np = n_kv & ~31, nblk = np >> 5, n_it = (nblk + 3) >> 2, np_pad = n_it * 128;
    const int ntail = n_kv - n8, nleft = n_kv - np;
    const int nq4 = r2 * hs / 4;
    if (aw) {
        // q of this kv head's heads (once per workgroup, through LDS), the lines of the V rows that hold the new column; zeros past the cache length
        if (tid < nq4) *(float4 *)(qs + tid * 4) = *(const float4 *)(a.q + (int64_t)kvh * r2 * hs + tid * 4);
        if (tid < PS_EXP2F_N) etab[tid] = ps_exp2f_tab[tid];
        const int vlim = pos0 & ~31, nkv4 = (n_kv + 3) & ~3; // (n_kv rounded up to 4 <= n_ctx: in bounds)
        if (tid < 32) {
            const int row = tid >> 3, col = vlim + 4 * (tid & 7);
            if (col < nkv4) *(float4 *)(vt + row * RS + col) = *(const float4 *)(vbase + (int64_t)row * a.n_ctx + col);
        }
        for (int col = nkv4 + 4 * tid; col < RS - 36; col += 4 * NT) {
#pragma unroll
            for (int k = 0; k < 4; k++) *(float4 *)(vt + k * RS + col) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) { // the new K row: the wave that owns the new position asks for its eight rows once more (the old ones are cache hits)
            const int sl = bx + ((uw >> 2) + SPP * ps) * G, j0 = sl * 32 + (uw & 3) * 8, j = j0 + p8;
            if (j0 <= pos0 && pos0 < j0 + 8) { // (uniform)
                const float *kr = kb + (int64_t)(j <= pos0 ? j : 0) * kvd;
                if (!KVS || j == pos0 || j0 == pos0) { // (KVS: the old rows were streamed past the caches -- only the new row, or the whole group when nobody has asked for it yet)
#pragma unroll
                    for (int m = 0; m < NV; m++) kf[ps][m] = *(const float4 *)(kr + m * 32);
                }
            }
        }
    }
    __syncthreads(); // B1
    mark(7);
    float *const xb = a.xchg + (size_t)kvh * 4 * XS;
    if (aw) { // ---- scores of this workgroup's positions (ggml_vec_dot_f32's chains and GGML_F32x8_REDUCE: attn_decode2_kernel)
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int sl = bx + ((uw >> 2) + SPP * ps) * G;
            if (sl * 32 + (uw & 3) * 8 < n_kv) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (g < r2) {
                        float x[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int m = 0; m < NV; m++) {
                            const float4 qv = *(const float4 *)(qs + g * hs + m * 32 + 4 * tq);
                            x[0] = __fmaf_rn(kf[ps][m].x, qv.x, x[0]);
                            x[1] = __fmaf_rn(kf[ps][m].y, qv.y, x[1]);
                            x[2] = __fmaf_rn(kf[ps][m].z, qv.z, x[2]);
                            x[3] = __fmaf_rn(kf[ps][m].w, qv.w, x[3]);
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) x[e] = __fadd_rn(x[e], dpp_f<0x104>(x[e]));
#pragma unroll
                        for (int e = 0; e < 4; e++) x[e] = __fadd_rn(x[e], dpp_f<0x102>(x[e]));
#pragma unroll
                        for (int e = 0; e < 4; e++) x[e] = __fadd_rn(x[e], dpp_f<0x101>(x[e]));
                        const float s = __fadd_rn(__fadd_rn(x[0], x[1]), __fadd_rn(x[2], x[3]));
                        if (tq == 0) sst[(((uw >> 2) + SPP * ps) * 4 + g) * 32 + (uw & 3) * 8 + p8] = s;
                    }
                }
            }
        }
    }
    __syncthreads(); // B2
    if (aw) {
#pragma unroll
        for (int rr = 0; rr < RMAX * 4 / AW; rr++) {
            const int row = uw + AW * rr, rd = row >> 2, gg = row & 3, sl = bx + rd * G;
            if (sl * 32 < n_kv && gg < r2 && lane < 32) qa_store_f(xb + (size_t)gg * XS + sl * 32 + lane, sst[row * 32 + lane]);
        }
    }
    mark(8); // scores stored
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // B3: every wave's score stores are drained
    if (tid == NT) qa_rendezvous(a.tick + kvh * 64, baseB, (unsigned)G, a.sync + 31); // ---- rendezvous B (the scores of the kv head)
    asm volatile("" ::: "memory");
    __syncthreads(); // B4
    mark(9);

    // ---- gather (plain loads of lines this launch has not touched), scale + mask, row maxima
    const int g = tid / LPH, t = tid % LPH;
    const bool hl = aw && g < r2;
    const float *xg = xb + (size_t)(hl ? g : 0) * XS;
    float sv[D2_TRIPS][8];
    float lmax = -INFINITY;
    if (aw) {
#pragma unroll
        for (int k = 0; k < D2_TRIPS; k++) {
            const int j0 = (t + LPH * k) * 8;
            const float *src = xg + ((hl && j0 < n_kv) ? j0 : 0);
            const float4 lo = *(const float4 *)src, hi = *(const float4 *)(src + 4);
            sv[k][0] = lo.x; sv[k][1] = lo.y; sv[k][2] = lo.z; sv[k][3] = lo.w; sv[k][4] = hi.x; sv[k][5] = hi.y; sv[k][6] = hi.z; sv[k][7] = hi.w;
        }
        if (a.kv_vis) { // hidden cache slots (KVCacheInterface::mask): -inf before the maximum
#pragma unroll
            for (int k = 0; k < D2_TRIPS; k++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int j = (t + LPH * k) * 8 + i;
                    if (hl && j < n_kv) {
                        const bool vis = j < pos0 ? a.kv_vis[j] != 0 : true;
                        float v = __fmul_rn(sv[k][i], a.scale);
                        v = __fadd_rn(v, vis ? 0.f : -INFINITY);
                        sv[k][i] = v;
                        lmax = fmaxf(lmax, v);
                    }
                }
        } else {
#pragma unroll
            for (int k = 0; k < D2_TRIPS; k++) {
                const int j0 = (t + LPH * k) * 8;
                if (hl && j0 < n_kv) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        sv[k][i] = __fmul_rn(sv[k][i], a.scale);
                        if (j0 + 8 <= n_kv || j0 + i < n_kv) lmax = fmaxf(lmax, sv[k][i]);
                    }
                }
            }
        }
        const float wm = wave_max_dpp(lmax);
        if (lane == 0) redf[wave] = wm;
#pragma unroll
        for (int k = 0; k < D2_TRIPS; k++)
            if (hl && (t + LPH * k) * 8 == n8) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (i < ntail) tails[g * 8 + i] = sv[k][i];
            }
    }
    __syncthreads(); // B5
    mark(10);
    if (aw) { // ---- e_j = exp(x_j - max), row sums in double (ggml.c:2831-2866)
        float mx = redf[(hl ? g : 0) * WPH];
#pragma unroll
        for (int w = 1; w < WPH; w++) mx = fmaxf(mx, redf[(hl ? g : 0) * WPH + w]);
        double rs = 0.0;
#pragma unroll
        for (int k = 0; k < D2_TRIPS; k++) {
            const int j0 = (t + LPH * k) * 8;
            if (!hl) continue;
            if (j0 + 8 <= n8) {
                ps_v_expf_n<8>(sv[k], mx);
                const float a0 = __fadd_rn(sv[k][4], sv[k][0]), a1 = __fadd_rn(sv[k][5], sv[k][1]), a2 = __fadd_rn(sv[k][6], sv[k][2]), a3 = __fadd_rn(sv[k][7], sv[k][3]);
                rs += (double)__fadd_rn(__fadd_rn(a0, a2), __fadd_rn(a1, a3));
                if (j0 + 8 <= np) {
                    *(float4 *)(pl + g * RS + j0)     = make_float4(sv[k][0], sv[k][1], sv[k][2], sv[k][3]);
                    *(float4 *)(pl + g * RS + j0 + 4) = make_float4(sv[k][4], sv[k][5], sv[k][6], sv[k][7]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) pleft[g * 32 + (j0 - np) + i] = sv[k][i];
                }
            }
            if (j0 >= np && j0 < np_pad) {
                *(float4 *)(pl + g * RS + j0)     = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4 *)(pl + g * RS + j0 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (hl && LPH - 1 - t < ntail) {
            const int idx = LPH - 1 - t;
            const float et = ps_expf_glibc(__fsub_rn(tails[g * 8 + idx], mx), etab);
            rs += (double)et;
            pleft[g * 32 + (n8 - np) + idx] = et;
        }
        const double sw = wave_sum_d_dpp(rs);
        if (lane == 0) redd[wave] = sw;
    }
    __syncthreads(); // B6
    mark(11);
    float lv[32], lp[32];
    if (aw) { // ---- V.p on v_mfma_f32_16x16x4_f32 as a k-ordered fma chain (attn_decode2_kernel): wave w owns chains 4w .. 4w + 3
        const int rl = lane & 15, kk = lane >> 4, ci = rl >> 2, rh = rl & 3;
        const bool bl = rh < r2;
        const int rb = bl ? rh : 0;
        double tots = redd[WPH * rb];
#pragma unroll
        for (int w = 1; w < WPH; w++) tots += redd[WPH * rb + w];
        const float inv = (float)(1.0 / tots);
        const float *va = vt + rh * RS + 4 * wave + ci + 32 * kk;
        const float *pb = pl + rb * RS + 4 * wave + ci + 32 * kk;
        ps_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float a0[4], b0[4], a1[4], b1[4];
        auto fetch = [&](float (&av)[4], float (&bv)[4], int s0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int sc = s0 + i < n_it ? s0 + i : n_it - 1;
                av[i] = va[128 * sc];
                bv[i] = pb[128 * sc];
            }
        };
        auto chain = [&](const float (&av)[4], const float (&bv)[4], int s0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float pj = __fmul_rn(bv[i], inv); // p_j = e_j * (float)(1/sum)
                const float b = __uint_as_float(__float_as_uint(pj) & ((bl && s0 + i < n_it) ? 0xffffffffu : 0u));
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b, acc, 0, 0, 0);
            }
        };
        if (n_it > 0) {
            fetch(a0, b0, 0);
            for (int s0 = 0; s0 < n_it; s0 += 8) {
                fetch(a1, b1, s0 + 4);
                __builtin_amdgcn_sched_barrier(0);
                chain(a0, b0, s0);
                if (s0 + 4 < n_it) {
                    fetch(a0, b0, s0 + 8);
                    __builtin_amdgcn_sched_barrier(0);
                    chain(a1, b1, s0 + 4);
                }
            }
        }
        if ((lane >> 4) == ((lane & 15) >> 2)) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) red[((4 * wave + (lane >> 4)) * 4 + rr) * 4 + (lane & 3)] = acc[rr];
        }
        if (tid >= NT - 16) { // the leftovers' operands meanwhile: v[jj], e[jj] * inv
            const int ch = (tid >> 2) & 3, hh = tid & 3, hb = hh < r2 ? hh : 0;
            double t2 = redd[WPH * hb];
#pragma unroll
            for (int w = 1; w < WPH; w++) t2 += redd[WPH * hb + w];
            const float inv2 = (float)(1.0 / t2);
#pragma unroll
            for (int jj = 0; jj < 32; jj++) {
                lv[jj] = vt[ch * RS + np + jj];
                lp[jj] = __fmul_rn(pleft[hb * 32 + jj], inv2);
            }
        }
    }
    __syncthreads(); // B7
    mark(12);
    if (aw && tid >= NT - 16) {
        const int ch = (tid >> 2) & 3, hh = tid & 3;
        float xc[32];
#pragma unroll
        for (int cc = 0; cc < 32; cc++) xc[cc] = red[(cc * 4 + ch) * 4 + hh];
        float t3[4];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) // GGML_F32x8_REDUCE (ggml.c:1354-1371)
            t3[cc] = __fadd_rn(__fadd_rn(__fadd_rn(xc[cc], xc[cc + 16]), __fadd_rn(xc[cc + 8], xc[cc + 24])),
                               __fadd_rn(__fadd_rn(xc[cc + 4], xc[cc + 20]), __fadd_rn(xc[cc + 12], xc[cc + 28])));
        float res = __fadd_rn(__fadd_rn(t3[0], t3[1]), __fadd_rn(t3[2], t3[3]));
#pragma unroll
        for (int jj = 0; jj < 32; jj++)
            if (jj < nleft) res = ps_dot_left(res, lv[jj], lp[jj], jj, nleft);
        if (hh < r2) a.att[((int64_t)kvh * r2 + hh) * hs + bx * 4 + ch] = res; // (write-through like the mat-vecs' rows: no difference, profiles/r06_out_wt_ab.txt)
    }
    if (dbg) { dbg[13] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

size_t qa_lds_bytes(const QAParams &p, int hs) {
    const int RS = ((p.a.n_ctx + 127) & ~127) + 36, RMAX = 16 / (hs / 32);
    const size_t small = (size_t)(512 + 128 + 16 + 32 + 4 * hs + RMAX * 128) * 4 + 32 * 8 + PS_EXP2F_N * 8;
    const size_t mv = (size_t)p.col_bytes + (size_t)p.rec_units * 64 * 16 + (size_t)8 * 32 * G4_HX + (size_t)3 * (p.split_q + 1) * 8 * 4;
    const size_t at = (size_t)4 * RS * 4;
    return small + at + (mv > at ? mv : at);
}

} // namespace

int g_qa_force = 0; // ps_hip_debug_set(7, v)

// Q / K / V mat-vec (RMSNorm + quantizer prologue, RoPE + KV append) + single-token attention in ONE launch.  false: not covered -- the caller issues the two launches.
bool psk_qkv_attn(hipStream_t st, int n_cu, const psk_gemv_args &g, int64_t K, const psl_attn_args &a) {
    static const bool off = getenv("PS_NO_QKV_ATTN") != nullptr; // (A/B switch for measurements)
    if (off) return false;
    const int hs = a.head_size, r2 = a.n_heads / a.n_kv_heads, G = hs / 4;
    if (g.n_w != 3 || !g.rope || g.rope_wi0 != 0 || g.pro != 1 || g.silu_pair || g.residual) return false;
    for (int i = 0; i < 3; i++) if (g.w[i]->dtype != PS_Q4_K || g.w[i]->K != K) return false;
    if (K % 1024 || K > 4096 || (K / 256) % 2) return false; // (one pair of activation tiles per producer)
    if (hs != 128 && hs != 64) return false;
    if (!a.xchg || !a.tick || !a.sync || a.tree || a.rope_pos || a.neox || a.k16 || a.v16 || r2 > 4 || a.n_ctx > QA_MAXCTX || a.n_dims > hs) return false;
    if (g.w[0]->N != (int64_t)a.n_heads * hs || g.w[1]->N != (int64_t)a.n_kv_heads * hs || g.w[2]->N != g.w[1]->N) return false;
    if (G * a.n_kv_heads > n_cu) return false; // every workgroup resident (one per CU: the LDS)
    // The launch has head_size / 4 workgroups per kv head: 256 for Llama-3.1-8B's shape, 128 for Llama-3.2-1B's (head size 64) -- half the chip for a mat-vec that the
    // unfused launch spreads over every CU: measured slower there (tools/gpu_fused_stress.py, profiles/r06_fused_stress.txt: 11.9 k vs 12.4 k tok/s on the 2-layer
    // 1B shape in Q4_K, against 4.84 k vs 4.77 k on the 4-layer 8B shape).  Taken only where its grid fills at least three quarters of the chip;
    // ps_hip_debug_set(7, 1) takes it wherever it is covered (the tests of the head-size-64 instance).
    extern int g_qa_force;
    if (!g_qa_force && 4 * G * a.n_kv_heads < 3 * n_cu) return false;
    QAParams p{};
    for (int i = 0; i < 3; i++) { p.qs[i] = g.w[i]->qs; p.aux[i] = g.w[i]->aux; p.bias[i] = g.bias[i]; }
    p.n_units = (int)(K / 256); p.K = (int)K; p.col_bytes = (int)psk_gemv_lds_col_bytes(PS_Q4_K, K);
    p.x = g.pro_x; p.nw = g.pro_norm_w; p.eps = g.pro_eps;
    p.nq = r2 * hs / 8; p.nk = hs / 8;
    const int T = p.nq + 2 * p.nk;
    p.split_q = T / G; p.split_r = T % G;
    if (p.split_q < 1) return false;
    const int max_units = (p.split_q + (p.split_r ? 1 : 0)) * p.n_units;
    p.rec_units = (max_units + 31) / 32 * 32;
    p.a = a;
    const size_t lds = qa_lds_bytes(p, hs);
    if (lds > 160 * 1024) return false;
    static unsigned long long attr = 0;
    if (ps_first_on_device(&attr)) {
        (void)hipFuncSetAttribute((const void *)qkv_attn_kernel<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)qkv_attn_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)qkv_attn_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)qkv_attn_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const dim3 grid((unsigned)(G * a.n_kv_heads));
    if (hs == 128) { if (a.kv_stream) hipLaunchKernelGGL((qkv_attn_kernel<4, 1>), grid, dim3(QA_THREADS), lds, st, p); else hipLaunchKernelGGL((qkv_attn_kernel<4, 0>), grid, dim3(QA_THREADS), lds, st, p); }
    else { if (a.kv_stream) hipLaunchKernelGGL((qkv_attn_kernel<2, 1>), grid, dim3(QA_THREADS), lds, st, p); else hipLaunchKernelGGL((qkv_attn_kernel<2, 0>), grid, dim3(QA_THREADS), lds, st, p); }
    return true;
}
