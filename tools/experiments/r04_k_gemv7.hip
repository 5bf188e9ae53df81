// Decode mat-vec, third generation (single activation column, Q4_K weights): gemv4's roles and arithmetic (producers turn 1 KiB units
// into the reference's integer partials, ONE chain wave runs the fp32 fma chains of ggml_vec_dot_q4_K_q8_K in unit order,
// libs/ggml/src/ggml-quants.c:7809-7873) with the weight stream taken OUT of the registers: every unit travels HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4: one 1 KiB unit of the lane-major repack per wave-instruction, non-temporal) into a ring of R slots per
// producer wave, so that R chunks (R x 36 KiB per CU) are requested at kernel entry -- before and during the RMSNorm / Q8_K prologue,
// which gemv4's two-deep register ring could not run ahead of (profiles/r02_gemv4_timeline.txt: 3-4 us of a 16-us launch with one
// 32 KiB chunk in flight).  MI355X_MICROARCH.md rows prefetch-credit / ldsdma-fill / nt-weights are the recipe.
//   * the CHAIN wave is the loader of the first two chunks of every producer (it has nothing else to do until records exist and it
//     may stall in the issue of 80 requests without holding up the prologue); it hands them over through the barriers the kernel
//     has anyway (its own counted s_waitcnt vmcnt, then s_barrier, then the producers' ds_reads);
//   * from then on a producer re-fills its own slot right after it has produced from it (5 requests: 4 units + their 4 x 128 B of
//     headers) and waits for its own requests with a counted vmcnt -- no extra synchronisation, R - 1 slots per wave in flight;
//   * producers have NO compiler-visible vector-memory operation in flight next to the DMA requests (hipcc neither counts an asm
//     request nor can it wait for one: its own waits would drain the ring), the activation row is loaded and waited for before the
//     first producer request, the epilogue operands are prefetched by a producer in the prologue;
//   * records shrink to what differs per lane: (float)sumi and (float)(mins . bsums) per lane and unit, d * yd / -dmin * yd once per
//     (unit, row) by the lane that expands the row's header -- 448 B per unit instead of 1 KiB, which is what lets ring, records,
//     expanded headers and the activation share 160 KiB of LDS.
// Epilogues: EPI 0 bias / residual, EPI 1 SiLU(gate)*up, EPI 2 RoPE + KV-cache append (QKV) -- gemv4's, unchanged.
#include "ps_gemv_dev.h"

namespace {
constexpr int G7_HX   = 48;          // expanded header of one (unit, row): sc16[4] | mins16[4] | d, dmin | pad
constexpr int G7_SLOT = 4096 + 512;  // ring slot: four consecutive 1 KiB units of one row group + their four 128-B header pieces
constexpr int G7_REC  = 448;         // record of one unit: s[64] floats | pr[8 rows][4] floats | {d * yd, -dmin * yd}[8 rows]

__device__ __forceinline__ unsigned g7_lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p; }

// One ring slot: 4 x 1 KiB of quants (lane l: bytes 16 l of each unit) + 512 B of headers (lanes 0..31), HBM -> LDS, no registers.
// M0 = LDS byte address of the destination (wave-uniform), the instruction offset applies to the global AND the LDS address,
// lane l lands at +16 l.  hipcc does not count these requests: every wait for them is a hand-written s_waitcnt vmcnt.
__device__ __forceinline__ void g7_dma_slot(const uint8_t *qg, const uint8_t *ag, const unsigned lds_slot, const unsigned lane16) {
    const uint8_t *qsrc = qg + lane16, *asrc = ag + (lane16 & 511u);
    const unsigned lds_hdr = lds_slot + 4096u;
    unsigned long long low32 = 0xffffffffull; // (lanes 0..31)
    unsigned keep;
    unsigned long long keepx;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %4\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off nt\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %2, off offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %2, off offset:3072 nt\n\t"
                 "s_mov_b32 m0, %5\n\t"
                 "s_mov_b64 %1, exec\n\t"
                 "s_and_b64 exec, exec, %6\n\t"
                 "s_nop 1\n\t"
                 "global_load_lds_dwordx4 %3, off nt\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(keepx)
                 : "v"(qsrc), "v"(asrc), "s"(lds_slot), "s"(lds_hdr), "s"(low32)
                 : "memory");
}
#define G7_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
constexpr int G7_OPS = 5; // requests per slot

// LDS image of the activation column for this kernel: quants (quad-major tiles), block scales, the 32-sums as int16 PAIRS
struct G7Act {
    const int *q32;
    const float *d;
    const uint32_t *bsp; // [unit][4]: {sum of block 2v, sum of block 2v + 1} (|sum| <= 4064 fits int16)
};

// A row's header of one super-block, expanded once per slot by the lane that holds it: sc16[j] = {scale[2j], scale[2j+1]} and
// mins16[v] = {min[2v], min[2v+1]} as int16 pairs (operands of v_dot2_i32_i16), d / dmin as fp32
__device__ __forceinline__ void g7_expand_header(const ps_u32x4 hc, char *dst, float &d, float &dmin) {
    const uint32_t sc03 = hc.y & 0x3f3f3f3fu, sc47 = (hc.w & 0x0f0f0f0fu) | (((hc.y >> 6) & 0x03030303u) << 4);
    const uint32_t mn03 = hc.z & 0x3f3f3f3fu, mn47 = ((hc.w >> 4) & 0x0f0f0f0fu) | (((hc.z >> 6) & 0x03030303u) << 4);
    *(uint4 *)dst = make_uint4(__builtin_amdgcn_perm(0u, sc03, 0x0c010c00u), __builtin_amdgcn_perm(0u, sc03, 0x0c030c02u),
                               __builtin_amdgcn_perm(0u, sc47, 0x0c010c00u), __builtin_amdgcn_perm(0u, sc47, 0x0c030c02u));
    *(uint4 *)(dst + 16) = make_uint4(__builtin_amdgcn_perm(0u, mn03, 0x0c010c00u), __builtin_amdgcn_perm(0u, mn03, 0x0c030c02u),
                                      __builtin_amdgcn_perm(0u, mn47, 0x0c010c00u), __builtin_amdgcn_perm(0u, mn47, 0x0c030c02u));
    d = ps_h2f((uint16_t)(hc.x & 0xffff));
    dmin = ps_h2f((uint16_t)(hc.x >> 16));
}
// one unit against the activation column: (float)sumi of accumulator lane u and (float)(mins . bsums) of acc_m lane u & 3.
// y0, y1: the lane's 32 quants of the super-block (quad-major: {g0, g1, g2, g3}, {g4 .. g7} of accumulator lane u), bs: its pair of 32-sums
// sc16, mp: the row's scale pairs and the mins pair of acc_m lane u & 3 from the expanded header
__device__ __forceinline__ void g7_unit(const ps_u32x4 q, const uint4 sc16, const uint32_t mp, const int4 y0, const int4 y1, const uint32_t bs, float &sf, float &prf) {
    constexpr uint32_t M = 0x0F0F0F0Fu;
    const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
    int dlo[4], dhi[4]; // the eight quad dots as plain v_dot4 (no zeroed accumulators)
    dot4x4(dlo, (int)(wq[0] & M), (int)(wq[1] & M), (int)(wq[2] & M), (int)(wq[3] & M), y0.x, y0.z, y1.x, y1.z);
    dot4x4(dhi, (int)((wq[0] >> 4) & M), (int)((wq[1] >> 4) & M), (int)((wq[2] >> 4) & M), (int)((wq[3] >> 4) & M), y0.y, y0.w, y1.y, y1.w);
    const uint32_t scv[4] = {sc16.x, sc16.y, sc16.z, sc16.w};
    // |dot4| <= 4*15*127 fits int16: {dl, dh} meet their scale pair in one v_dot2_i32_i16 (exact); two chains of two, then one add
    int s0 = dot2_i16(__builtin_amdgcn_perm((uint32_t)dhi[0], (uint32_t)dlo[0], 0x05040100u), scv[0], 0);
    int s1 = dot2_i16(__builtin_amdgcn_perm((uint32_t)dhi[1], (uint32_t)dlo[1], 0x05040100u), scv[1], 0);
    s0 = dot2_i16(__builtin_amdgcn_perm((uint32_t)dhi[2], (uint32_t)dlo[2], 0x05040100u), scv[2], s0);
    s1 = dot2_i16(__builtin_amdgcn_perm((uint32_t)dhi[3], (uint32_t)dlo[3], 0x05040100u), scv[3], s1);
    sf  = (float)(s0 + s1);
    prf = (float)dot2_i16(mp, bs, 0); // mins[2v] * q8sum[2v] + mins[2v+1] * q8sum[2v+1]   (ggml-quants.c:7831-7834; exact in int32)
}

// Q8_K quantization of one 256-element tile held 4 values per lane (quantize_row_q8_K_ref, ggml-quants.c:3799-3835): g4_quantize_tile
// with the 32-sums stored as int16 (the pair of a sub-block pair is one dword for g7_unit)
__device__ __forceinline__ void g7_quantize_tile(const float v[4], const int e, const int t, int8_t *qs, float *d, int16_t *bs16, const bool live) {
    int q[4];
    const float am   = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    const float amax = wave_max_dpp(am);
    const unsigned long long hits = __ballot(am == amax);
    const float mine = fabsf(v[0]) == amax ? v[0] : fabsf(v[1]) == amax ? v[1] : fabsf(v[2]) == amax ? v[2] : v[3];
    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), __ffsll((long long)hits) - 1));
    const bool zero = amax == 0.f;
    const float iscale = zero ? 0.f : __fdiv_rn(-127.f, mx);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = min(127, __float2int_rn(__fmul_rn(iscale, v[i])));
    const float dd = zero ? 0.f : __fdiv_rn(1.0f, iscale);
    int s = q[0] + q[1] + q[2] + q[3];
    s += dpp_i<0xB1>(s); s += dpp_i<0x4E>(s); s += dpp_i<0x141>(s);
    if (live) {
        const int dw = (e >> 2) & 63, eq = (e & ~255) + (((dw & 7) << 3) | (dw >> 3)) * 4;
        *(uint32_t *)(qs + eq) = (uint32_t)(q[0] & 0xff) | ((uint32_t)(q[1] & 0xff) << 8) | ((uint32_t)(q[2] & 0xff) << 16) | ((uint32_t)(q[3] & 0xff) << 24);
        d[t] = dd;
        bs16[e >> 5] = (int16_t)s;
    }
}

struct G7Mat {
    const uint8_t *qs, *aux;
    float *out;
    const float *bias;
    int64_t N;
    int n_groups;
};
struct G7Params {
    G7Mat w[3];
    int n_w, n_units, n_tasks;  // tasks: row groups (EPI 0 / 2) or gate/up row-group pairs (EPI 1)
    int split_q, split_r;       // tasks per workgroup = split_q (+1 for the first split_r workgroups)
    int K;
    const float *residual;
    const float *x, *nw;        // PRO 1: rmsnorm(x, nw, eps) then quantize;  PRO 2: quantize(x)
    float eps;
    const int8_t *aq;           // PRO 0: activation already quantized
    const float *ad;
    const int16_t *abs16;
    unsigned long long *dbg;
    psk_rope_kv rope;           // EPI 2
    int rope_wi0;               // EPI 2: w[0]'s place in the Q / K / V triple
};

// LDS carve-up (one dynamic array: ring first, so that its slots sit at fixed offsets)
template <int NW, int R> struct G7Lds {
    static constexpr int UPB = NW * 4;
    static constexpr size_t ring = 0, ring_bytes = (size_t)NW * R * G7_SLOT;
    static constexpr size_t recs = ring + ring_bytes, recs_bytes = (size_t)2 * UPB * G7_REC;
    static constexpr size_t hscr = recs + recs_bytes, hscr_bytes = (size_t)NW * 32 * G7_HX;
    static constexpr size_t red = hscr + hscr_bytes, red_bytes = 16 * 8;
    static constexpr size_t etab = red + red_bytes, etab_bytes = PS_EXP2F_N * 8;
    static constexpr size_t act = etab + etab_bytes; // lq[K] | ld[n_units] | bsp[n_units * 4] | ep[3][(split_q + 1) * 8]
    static size_t total(int K, int split_q) { return act + (size_t)K + (size_t)(K / 256) * 4 + (size_t)(K / 256) * 16 + (size_t)3 * (split_q + 1) * 8 * 4; }
};

// NW producer waves, R ring slots per producer, TPW activation tiles per producer, EPI / PRO as gemv4.
// YS: the activation operands of a producer's units live in REGISTERS (YS = 1: a wave meets the same four super-blocks in every chunk,
// UPB % tot == 0 -- every K = 4096 launch; YS = 2: two alternating sets, 2 UPB % tot == 0 -- K = 14336 with 7 producers; 0: read from
// LDS per unit).  All eight rows of a unit multiply the SAME 32 activation bytes per accumulator lane: fetched per unit that is two
// ds_read_b128 + one ds_read_b32 per lane -- 2 KiB through the LDS pipe for every 1 KiB of weights, the largest share of its traffic.
template <int NW, int R, int TPW, int YS, int EPI, int PRO>
__global__ __launch_bounds__((NW + 1) * 64) void gemv7_kernel(const G7Params p) {
    constexpr int WT = PS_Q4_K;
    using TR = WTraits<WT>;
    using LY = G7Lds<NW, R>;
    constexpr int UPW = 4, UPB = NW * UPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K, n_units = p.n_units;
    double *red        = (double *)(smem + LY::red);
    uint64_t *exp_tab  = (uint64_t *)(smem + LY::etab);
    int8_t *lq         = (int8_t *)(smem + LY::act);
    float *ld          = (float *)(lq + K);
    uint32_t *lbp      = (uint32_t *)(ld + n_units);
    float *epA         = (float *)(lbp + n_units * 4);
    const int ep_n     = (p.split_q + 1) * 8;
    float *const epB = epA + ep_n, *const epC = epB + ep_n;
    char *recs         = smem + LY::recs; // [2][UPB] records
    const unsigned lds0 = g7_lds_addr(smem);
    G7Act A;
    A.q32 = (const int *)lq; A.d = ld; A.bsp = lbp;
    const int r = lane >> 3, u = lane & 7;
    const uint32_t lane16 = (uint32_t)lane * 16u;

    const int tot = (EPI == 1) ? 2 * n_units : n_units; // stream units per task (EPI 1: gate units then up units)
    const int t0  = (int)blockIdx.x * p.split_q + min((int)blockIdx.x, p.split_r);
    const int nt  = p.split_q + ((int)blockIdx.x < p.split_r ? 1 : 0);
    const int s_end    = nt * tot;                  // stream units of this workgroup (a multiple of 4)
    const int n_chunks = (s_end + UPB - 1) / UPB;
    unsigned long long *const dbg = (p.dbg && blockIdx.x < 1024 && lane == 0 && (wave == 0 || wave == NW)) ? p.dbg + ((size_t)blockIdx.x * 2 + (wave == NW)) * 32 : nullptr;
    int dbg_n = 0;
    auto mark = [&]() { if (dbg && dbg_n < 28) dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); };
    mark(); // 0: entry
    if (dbg) dbg[29] = __builtin_amdgcn_s_memrealtime();

    // global addresses of the slot that starts at (local task tl, unit un of the task's stream)
    auto slot_src = [&](int tl, int un, const uint8_t *&qg, const uint8_t *&ag) {
        int grp = t0 + tl, ul = un;
        const uint8_t *qb = p.w[0].qs, *ab = p.w[0].aux;
        if (EPI == 1) {
            if (ul >= n_units) { ul -= n_units; qb = p.w[1].qs; ab = p.w[1].aux; }
        } else if (p.n_w > 1 && grp >= p.w[0].n_groups) {
            grp -= p.w[0].n_groups; qb = p.w[1].qs; ab = p.w[1].aux;
            if (p.n_w > 2 && grp >= p.w[1].n_groups) { grp -= p.w[1].n_groups; qb = p.w[2].qs; ab = p.w[2].aux; }
        }
        const uint32_t idx = (uint32_t)(grp * n_units + ul);
        qg = qb + ((uint64_t)idx << 10);
        ag = ab + ((uint64_t)idx << 7);
    };
    const int step_t = UPB / tot, step_u = UPB % tot; // a wave's slot moves one chunk
    auto advance = [&](int &tl, int &un) {
        tl += step_t; un += step_u;
        if (un >= tot) { un -= tot; tl++; }
    };

    if (wave < NW) { // ------------------------------------------------------------------ producers
        // 1. the activation row (tile t -> wave t % NW); wave 0 also fetches what the chain wave's epilogue will read
        float4 xv[TPW], wv[TPW];
        if (PRO != 0) ps_qrow_load<(PRO == 1 ? 1 : 0), TPW>(p.x, p.nw, K, xv, wv, NW);
        if (wave == 0) {
            if (EPI == 1) {
                if (lane < PS_EXP2F_N) exp_tab[lane] = ps_exp2f_tab[lane];
            } else {
                int rpos = 0;
                if (EPI == 2) { const int kv_pos = p.rope.state->pos0; rpos = p.rope.rope_pos ? p.rope.rope_pos[0] : kv_pos; }
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
        }
        mark(); // 1: loads issued
        // 2. activation -> LDS (the chain wave joins the barriers)
        if (PRO == 0) {
            for (int i = threadIdx.x; i < K / 4; i += NW * 64) { // (quad-major tiles, as the quantizer writes them)
                const int dw = i & 63;
                ((int *)lq)[(i & ~63) + (((dw & 7) << 3) | (dw >> 3))] = ((const int *)p.aq)[i];
            }
            for (int i = threadIdx.x; i < n_units; i += NW * 64) ld[i] = p.ad[i];
            for (int i = threadIdx.x; i < K / 32; i += NW * 64) ((int16_t *)lbp)[i] = (int16_t)((int)p.abs16[2 * i] + (int)p.abs16[2 * i + 1]);
            __syncthreads();
        } else {
            auto pmark = [&](int k) { if (dbg && wave == 0) dbg[32 + k] = __builtin_amdgcn_s_memtime(); };
            float scale = 1.0f;
            if (PRO == 1) { // RMSNorm: ggml.c:12667-12720, double sum of squares, scale = 1/sqrtf(mean + eps), y = x * (w * scale)
                double ss = 0.0;
#pragma unroll
                for (int i = 0; i < TPW; i++) {
                    ss += (double)__fmul_rn(xv[i].x, xv[i].x);
                    ss += (double)__fmul_rn(xv[i].y, xv[i].y);
                    ss += (double)__fmul_rn(xv[i].z, xv[i].z);
                    ss += (double)__fmul_rn(xv[i].w, xv[i].w);
                }
                pmark(24); // the row has arrived
                ss = wave_sum_d_dpp(ss);
                if (lane == 0) red[wave] = ss;
                __syncthreads();
                pmark(25);
                double tot_ss = 0.0;
#pragma unroll
                for (int i = 0; i <= NW; i++) tot_ss += red[i];
                const float mean = (float)(tot_ss / (double)K);
                scale            = __fdiv_rn(1.0f, sqrtf(__fadd_rn(mean, p.eps)));
                pmark(26);
            }
#pragma unroll
            for (int i = 0; i < TPW; i++) {
                const int t = wave + i * NW;
                const bool live = t < n_units; // wave-uniform; a dead tile runs on zeros and stores nothing
                float v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
                if (PRO == 1) {
                    v[0] = __fmul_rn(v[0], __fmul_rn(wv[i].x, scale));
                    v[1] = __fmul_rn(v[1], __fmul_rn(wv[i].y, scale));
                    v[2] = __fmul_rn(v[2], __fmul_rn(wv[i].z, scale));
                    v[3] = __fmul_rn(v[3], __fmul_rn(wv[i].w, scale));
                }
                g7_quantize_tile(v, t * 256 + lane * 4, t, lq, ld, (int16_t *)lbp, live);
            }
            pmark(27);
            __syncthreads();
        }
        mark(); // 2: activation in LDS, chunk 0 landed (the chain wave waited for it)
        // 3. own slots: chunk c sits in ring slot c % R of this wave.  Chunks 0, 1 were requested by the chain wave; this wave
        //    requests 2 .. R-1 now and chunk c + R when it has produced chunk c.  my_n: chunks in which this wave has a slot.
        const int my_n = (s_end - wave * UPW + UPB - 1) / UPB; // (<= 0: none)
        int tP = 0, uP = wave * UPW, tI = 0, uI = wave * UPW; // cursors: slot being produced / slot to request next
        while (uP >= tot) { uP -= tot; tP++; }
        tI = tP; uI = uP;
        int cI = 0;                                           // chunk index of the request cursor
        auto request = [&]() { // the slot of chunk cI into ring slot cI % R, when it exists
            if (cI < my_n) {
                const uint8_t *qg, *ag;
                slot_src(tI, uI, qg, ag);
                g7_dma_slot(qg, ag, lds0 + (unsigned)(LY::ring + (size_t)(wave * R + cI % R) * G7_SLOT), lane16);
            }
            cI++;
            advance(tI, uI);
        };
        cI = 2; advance(tI, uI); advance(tI, uI); // chunks 0, 1: the chain wave's
#pragma unroll
        for (int k = 2; k < R; k++) request();
        char *const hs = smem + LY::hscr + (size_t)wave * 32 * G7_HX;
        // the activation operands of this wave's units, register-resident (YS sets of four super-blocks)
        constexpr int YN = YS ? YS : 1;
        int4 Y0[YN][UPW], Y1[YN][UPW];
        uint32_t BS[YN][UPW];
        float YD[YN];
        if (YS) {
            int ts = tP, us = uP;
#pragma unroll
            for (int k = 0; k < YN; k++) {
                const int ul = (EPI == 1 && us >= n_units) ? us - n_units : us;
#pragma unroll
                for (int i = 0; i < UPW; i++) {
                    Y0[k][i] = *(const int4 *)(A.q32 + (ul + i) * 64 + u * 8);
                    Y1[k][i] = *(const int4 *)(A.q32 + (ul + i) * 64 + u * 8 + 4);
                    BS[k][i] = A.bsp[(ul + i) * 4 + (u & 3)];
                }
                YD[k] = A.d[ul + ((lane >> 3) & 3)];
                advance(ts, us);
            }
        }
        auto body = [&](auto setc, const int c) {
            constexpr int SET = decltype(setc)::value;
            if (c < my_n) {
                if (dbg && c >= 6) dbg_n = 28; // (the timeline holds six chunks of four marks)
                // own requests behind chunk c's: chunks c+1 .. min(c + R - 1, my_n - 1), and none of them when c < 2 was the chain wave's
                // (then this wave's queue holds only later chunks: the wait below is already satisfied)
                const int later = min(R - 1, my_n - 1 - c);
                if (R == 3 && later >= 2) G7_WAIT_VM(10);
                else if (later >= 1) G7_WAIT_VM(5);
                else G7_WAIT_VM(0);
                mark(); // 3 + 4c: the slot has landed
                const char *slot = smem + LY::ring + (size_t)(wave * R + c % R) * G7_SLOT;
                const int ul = (EPI == 1 && uP >= n_units) ? uP - n_units : uP;
                char *rb = recs + (size_t)((c & 1) * UPB + wave * UPW) * G7_REC;
                if (lane < 32) { // lane l holds the header of (unit l >> 3, row l & 7): expand it, leave the chain's two factors
                    const ps_u32x4 hc = *(const ps_u32x4 *)(slot + 4096 + lane * 16);
                    float d, dmin;
                    g7_expand_header(hc, hs + lane * G7_HX, d, dmin);
                    const float yd = YS ? YD[SET] : A.d[ul + (lane >> 3)];
                    *(float2 *)(rb + (size_t)(lane >> 3) * G7_REC + 384 + (lane & 7) * 8) = make_float2(__fmul_rn(yd, d), __fmul_rn(-yd, dmin));
                }
                // every LDS operand of the slot's four units is requested up front (one exposed round trip per slot instead of one per unit:
                // the units themselves are dependent chains of ~33 instructions with nothing to overlap a wait with)
                ps_u32x4 q[UPW];
                uint4 sc[UPW];
                uint32_t mp[UPW];
#pragma unroll
                for (int i = 0; i < UPW; i++) {
                    q[i]  = *(const ps_u32x4 *)(slot + i * 1024 + lane16);
                    sc[i] = *(const uint4 *)(hs + (i * 8 + r) * G7_HX);
                    mp[i] = *(const uint32_t *)(hs + (i * 8 + r) * G7_HX + 16 + (u & 3) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < UPW; i++) {
                    float sf, prf;
                    if (YS) {
                        g7_unit(q[i], sc[i], mp[i], Y0[SET][i], Y1[SET][i], BS[SET][i], sf, prf);
                    } else {
                        const int4 y0 = *(const int4 *)(A.q32 + (ul + i) * 64 + u * 8), y1 = *(const int4 *)(A.q32 + (ul + i) * 64 + u * 8 + 4);
                        g7_unit(q[i], sc[i], mp[i], y0, y1, A.bsp[(ul + i) * 4 + (u & 3)], sf, prf);
                    }
                    *(float *)(rb + (size_t)i * G7_REC + lane * 4) = sf;
                    *(float *)(rb + (size_t)i * G7_REC + 256 + (r * 4 + (u & 3)) * 4) = prf; // (lanes u, u + 4: the same value)
                    __builtin_amdgcn_sched_barrier(0);
                }
                mark(); // 4 + 4c: chunk produced
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slot has been read: it may be overwritten
            }
            advance(tP, uP);
            request();
            mark(); // 5 + 4c: next request out
            __syncthreads();
            mark(); // 6 + 4c: barrier passed
        };
        for (int c = 0; c < n_chunks; c += YN) {
            body(std::integral_constant<int, 0>{}, c);
            if constexpr (YN == 2) { if (c + 1 < n_chunks) body(std::integral_constant<int, 1>{}, c + 1); }
        }
    } else { // ------------------------------------------------------------------------- chain wave
        // loader of chunks 0 and 1: producer w's slot of chunk c starts at stream unit c * UPB + 4 w
        int tl = 0, un = 0;
        auto burst = [&](int c) {
            for (int w = 0; w < NW; w++) {
                if (c * UPB + w * UPW < s_end) {
                    const uint8_t *qg, *ag;
                    slot_src(tl, un, qg, ag);
                    g7_dma_slot(qg, ag, lds0 + (unsigned)(LY::ring + (size_t)(w * R + c) * G7_SLOT), lane16);
                }
                un += UPW;
                if (un >= tot) { un -= tot; tl++; }
            }
        };
        burst(0);
        if (PRO == 1) { // the sum-of-squares exchange: nobody waits for this wave there
            if (lane == 0) red[wave] = 0.0;
            __syncthreads();
        }
        if (R >= 2) burst(1);
        // chunk 0 has landed when at most chunk 1's requests are outstanding (a chunk that is not full: wait for everything)
        if (R >= 2 && s_end >= 2 * UPB) { if (NW == 8) G7_WAIT_VM(40); else if (NW == 7) G7_WAIT_VM(35); else G7_WAIT_VM(0); }
        else G7_WAIT_VM(0);
        int kv_pos = 0;
        if (EPI == 2) kv_pos = p.rope.state->pos0;
        __syncthreads();
        mark(); // 1: activation in LDS
        __builtin_amdgcn_s_setprio(3); // one wave serves NW producers: it gets the issue slots first
        float acc0 = 0.f, acc1 = 0.f, accm = 0.f, ygate = 0.f;
        tl = 0; un = 0; // local task, units of it already chained
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
                const psk_rope_kv &R_ = p.rope;
                const int role = wi + p.rope_wi0; // 0 q, 1 k, 2 v
                if (u == 0 && row < Nw) {
                    if (role == 2) {
                        R_.v_cache[row * R_.n_ctx + kv_pos] = v;
                        if (R_.v16) R_.v16[(int64_t)kv_pos * R_.kv_dim + row] = (_Float16)v;
                    } else {
                        const int e = (int)(row % R_.head_size);
                        float res = v;
                        if (e < R_.n_dims) {
                            const float c = ea, sn = eb;
                            const float x0 = (e & 1) ? vp : v, x1 = (e & 1) ? v : vp;
                            res = ps_rope_one(x0, x1, c, sn, (e & 1) != 0);
                        }
                        if (role == 0) o[row] = res; else { R_.k_cache[(int64_t)kv_pos * R_.kv_dim + row] = res; if (R_.k16) R_.k16[(int64_t)kv_pos * R_.kv_dim + row] = (_Float16)res; }
                    }
                }
            } else if (u == 0 && row < Nw) {
                if (EPI == 1) {
                    o[row] = g4_silu_mul(ygate, y, exp_tab);
                } else {
                    float v = y;
                    if (b) v = __fadd_rn(v, ec);
                    if (p.residual && wi == 0) v = __fadd_rn(ea, v);
                    o[row] = v;
                }
            }
            acc0 = 0.f; acc1 = 0.f; accm = 0.f;
            un = 0;
            tl++;
        };
        auto batch = [&](auto nconst, const char *rb, const int k0) { // N records in one LDS round trip, then the two fma chains
            constexpr int N = decltype(nconst)::value;
            float rs[N], rp[N];
            float2 rd[N];
#pragma unroll
            for (int k = 0; k < N; k++) {
                const char *rk = rb + (size_t)(k0 + k) * G7_REC;
                rs[k] = *(const float *)(rk + lane * 4);
                rp[k] = *(const float *)(rk + 256 + (r * 4 + (u & 3)) * 4);
                rd[k] = *(const float2 *)(rk + 384 + r * 8);
            }
#pragma unroll
            for (int k = 0; k < N; k++) {
                acc0 = __fmaf_rn(rd[k].x, rs[k], acc0);
                accm = __fmaf_rn(rd[k].y, rp[k], accm); // lanes u >= 4: not an acc_m lane, never read
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int c = 0; c < n_chunks; c++) {
            mark(); // 2, 4, ...: previous chunk chained, waiting
            if (c == 0) G7_WAIT_VM(0); // chunk 1 has landed before the barrier that opens its produce phase (nothing else of this wave is in flight)
            __syncthreads();
            mark(); // 3, 5, ...: chunk c handed over
            const char *rb = recs + (size_t)(c & 1) * UPB * G7_REC;
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

template <int NW, int R, int TPW, int YS, int EPI, int PRO>
int launch_g7(hipStream_t st, int grid, const G7Params &p) {
    const size_t smem = G7Lds<NW, R>::total(p.K, p.split_q);
    if (smem > 160 * 1024) return -1;
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) {
        if (hipFuncSetAttribute((const void *)gemv7_kernel<NW, R, TPW, YS, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
    }
    psk_note_kernel("gemv7_kernel<%d, %d, %d, %d, %d, %d>", NW, R, TPW, YS, EPI, PRO);
    hipLaunchKernelGGL((gemv7_kernel<NW, R, TPW, YS, EPI, PRO>), dim3((unsigned)grid), dim3((NW + 1) * 64), smem, st, p);
    return 0;
}

// KC: row-length class (tiles of 256 per row <= 16 << KC)
template <int NW, int R, int YS, int KC>
int launch_g7_ep(hipStream_t st, int grid, const G7Params &p, int epi, int pro) {
    constexpr int TPW = ((16 << KC) + NW - 1) / NW;
    if (epi == 2) { if (pro != 1) return -1; return launch_g7<NW, R, TPW, YS, 2, 1>(st, grid, p); }
    if (epi == 1) {
        if (pro == 1) return launch_g7<NW, R, TPW, YS, 1, 1>(st, grid, p);
        if (pro == 0) return launch_g7<NW, R, TPW, YS, 1, 0>(st, grid, p);
        return -1;
    }
    if (pro == 0) return launch_g7<NW, R, TPW, YS, 0, 0>(st, grid, p);
    if (pro == 1) return launch_g7<NW, R, TPW, YS, 0, 1>(st, grid, p);
    return launch_g7<NW, R, TPW, YS, 0, 2>(st, grid, p);
}
template <int NW, int R, int YS>
int launch_g7_kc(hipStream_t st, int grid, const G7Params &p, int epi, int pro) {
    if (p.n_units <= 16) return launch_g7_ep<NW, R, YS, 0>(st, grid, p, epi, pro);
    if (p.n_units <= 64) return launch_g7_ep<NW, R, YS, 2>(st, grid, p, epi, pro);
    return -1;
}
template <int NW, int YS>
int launch_g7_r(hipStream_t st, int grid, const G7Params &p, int epi, int pro, int r_max) {
    if (r_max >= 3) { const int rc = launch_g7_kc<NW, 3, YS>(st, grid, p, epi, pro); if (rc != -1) return rc; } // (-1: does not fit the LDS)
    return launch_g7_kc<NW, 2, YS>(st, grid, p, epi, pro);
}

} // namespace

// Single-column Q4_K mat-vec on the LDS-DMA ring.  cfg (ps_hip_debug_set(1, 20 + v), tools/g4_variants.py): bit 0: two ring slots per
// producer instead of three, bit 1: activation operands from LDS per unit instead of registers.  Returns -1 when the launch is not covered (the caller goes on to gemv4).
int psk_gemv7(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K, int cfg) {
    if (a.n_w < 1 || a.n_w > 3 || K % 1024 != 0 || K > 16384) return -1;
    G7Params p{};
    int groups_total = 0;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != PS_Q4_K || a.w[i]->K != K) return -1;
        const int ng = (int)((a.w[i]->N + 7) / 8);
        p.w[i] = G7Mat{a.w[i]->qs, a.w[i]->aux, a.out[i], a.bias[i], a.w[i]->N, ng};
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
    int grid = p.n_tasks < n_cu ? p.n_tasks : n_cu;
    if (grid < 1) return -1;
    p.split_q = p.n_tasks / grid; p.split_r = p.n_tasks % grid;
    p.dbg = psk_gemv_dbg_buf(epi, a.pro);
    const bool seven = p.n_units % 7 == 0; // rows of a multiple of 7 units: 7 producers, a chunk is half a row
    const int r_max = (cfg & 1) ? 2 : 3;
    const int tot = (epi == 1 ? 2 : 1) * p.n_units, upb = (seven ? 7 : 8) * 4;
    const int ys = (cfg & 2) ? 0 : (upb % tot == 0 ? 1 : ((2 * upb) % tot == 0 ? 2 : 0)); // register-resident activation operands where a wave's units repeat
    if (seven) return ys == 2 ? launch_g7_r<7, 2>(st, grid, p, epi, a.pro, r_max) : (ys == 1 ? launch_g7_r<7, 1>(st, grid, p, epi, a.pro, r_max) : launch_g7_r<7, 0>(st, grid, p, epi, a.pro, r_max));
    return ys == 2 ? launch_g7_r<8, 2>(st, grid, p, epi, a.pro, r_max) : (ys == 1 ? launch_g7_r<8, 1>(st, grid, p, epi, a.pro, r_max) : launch_g7_r<8, 0>(st, grid, p, epi, a.pro, r_max));
}
