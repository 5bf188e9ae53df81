// KV-cache attention for the fused model path (gfx950) — bit-exact with the reference's F32 path.
//
// Fuses what NormAttention::build emits after the QKV mat-muls
// (src/model/module/norm_attention.cpp:72-151): ROPE x2, TRANSPOSE, VIEW+COPY x2 (KV append), PERMUTE,
// K-view MAT_MUL, GET_MASK (src/executor/executor.cpp:210-224), SOFTMAX_EXT, V-view MAT_MUL, PERMUTE+CONT —
// 13 graph ops — into three launches with no intermediate layout shuffles:
//   rope_append   rotate q in place, rotate k into its K-cache row, scatter v into the transposed V cache
//   attn_scores      s = K·q       (ggml_vec_dot_f32, libs/ggml/src/ggml.c:2092-2133)
//   attn_softmax_pv  scale + mask + softmax (ggml.c:14889-14925, ggml_vec_soft_max_f32 :2814-2863) held in LDS,
//                    then out = V·p (ggml_vec_dot_f32 again, rows of the transposed V cache)
// KV layout is the reference's: K [n_ctx][kv_dim], V [kv_dim][n_ctx] FP32 (backend/ggml/ggml_kv_cache.cpp:48-57).
//
// Exact order: ggml_vec_dot_f32's AVX build keeps 4 accumulators x 8 lanes = 32 fp32 chains over elements
// e = 32*i + c (c = 0..31), reduces them as (acc0+acc2, acc1+acc3), (+), low/high 128-bit halves, two hadds
// (GGML_F32x8_REDUCE, ggml.c:1354-1371), and adds the n % 32 leftovers one by one.  Here 32 GPU lanes own the
// 32 chains of one dot product (coalesced 128-B reads) and the reduction is the xor-16, 8, 4, 1, 2 butterfly.
// All position-dependent values come from a device-resident ps_step_state so a captured hipGraph replays.
#include "ps_dev.h"
#include "ps_expf.h"
#include "ps_ops.h"
#include "ps_quant_dev.h"

namespace {

// agent-scope relaxed accesses (global_load / global_store ... sc1): what one workgroup hands to another INSIDE a launch
__device__ __forceinline__ float coh_load_f(const float *p) { return __uint_as_float(__hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void coh_store_f(float *p, float v) { __hip_atomic_store((uint32_t *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (pos0, bs) of the forward: from the launch arguments when the host knew them at enqueue time (eager batches), else from the device-resident state
__device__ __forceinline__ int att_pos0(const psl_attn_args &a) { return a.bs_host > 0 ? a.n_kv_host - a.bs_host : a.state->pos0; }
__device__ __forceinline__ int att_bs(const psl_attn_args &a) { return a.bs_host > 0 ? a.bs_host : a.state->bs; }

// ---------------------------------------------------------------- rope + KV append
__global__ void rope_append_kernel(psl_attn_args a, int bs) {
    const int hs = a.head_size, dim = a.n_heads * hs, kvd = a.n_kv_heads * hs, half = a.n_dims / 2;
    const int pos0 = att_pos0(a);
    const int64_t nq = (int64_t)bs * a.n_heads * (hs / 2), nk = (int64_t)bs * a.n_kv_heads * (hs / 2), nv = (int64_t)bs * kvd;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nq + nk + nv; o += (int64_t)gridDim.x * blockDim.x) {
        if (o < nq + nk) {
            const bool isq = o < nq;
            const int64_t oo = isq ? o : o - nq;
            const int nh = isq ? a.n_heads : a.n_kv_heads;
            const int pi = (int)(oo % (hs / 2));
            const int h = (int)((oo / (hs / 2)) % nh), i = (int)(oo / ((int64_t)(hs / 2) * nh));
            const int p = pos0 + i, i0 = 2 * pi;                 // p: cache slot
            const int rp = a.rope_pos ? a.rope_pos[i] : p;      // RoPE position
            const float *src = isq ? a.q + (int64_t)i * dim + h * hs : a.k + (int64_t)i * kvd + h * hs;
            float *dst = isq ? a.q + (int64_t)i * dim + h * hs : a.k_cache + (int64_t)p * kvd + h * hs;
            if (i0 >= a.n_dims) {
                dst[i0] = src[i0]; dst[i0 + 1] = src[i0 + 1];
                if (!isq && a.k16) { a.k16[(int64_t)p * kvd + h * hs + i0] = (_Float16)src[i0]; a.k16[(int64_t)p * kvd + h * hs + i0 + 1] = (_Float16)src[i0 + 1]; }
                continue;
            }
            const float c = a.rope_table[(int64_t)rp * hs + i0], s = a.rope_table[(int64_t)rp * hs + i0 + 1];
            const int ia = a.neox ? pi : i0, ib = a.neox ? pi + half : i0 + 1;
            const float x0 = src[ia], x1 = src[ib];
            float ra, rb;
            ps_rope_pair(x0, x1, c, s, ra, rb);
            dst[ia] = ra;
            dst[ib] = rb;
            if (!isq && a.k16) { a.k16[(int64_t)p * kvd + h * hs + ia] = (_Float16)ra; a.k16[(int64_t)p * kvd + h * hs + ib] = (_Float16)rb; }
        } else {
            const int64_t oo = o - nq - nk;
            const int d = (int)(oo % kvd), i = (int)(oo / kvd);
            a.v_cache[(int64_t)d * a.n_ctx + pos0 + i] = a.v[(int64_t)i * kvd + d];
            if (a.v16) a.v16[(int64_t)(pos0 + i) * kvd + d] = (_Float16)a.v[(int64_t)i * kvd + d];
        }
    }
}

// ---------------------------------------------------------------- scores: s[i][h][j] = K[j][kvh] · q[i][h]
// grid (n_ctx/32, n_kv_heads, bs); a half-wave (32 lanes) per position, 8 half-waves x 4 rounds = 32 positions.
constexpr int R2MAX = 8;
template <int NV> // head_size / 32
__global__ __launch_bounds__(256) void attn_scores_kernel(psl_attn_args a) {
    const int hs = NV * 32, dim = a.n_heads * hs, kvd = a.n_kv_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int c = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int kvh = blockIdx.y, i = blockIdx.z;
    const int n_kv = att_pos0(a) + att_bs(a);
    const int j0 = blockIdx.x * 32;
    if (j0 >= n_kv) return;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x; // timeline (tools/gpu_attn_timeline.py, key 40)
    unsigned long long *const dbg = (PS_TL(a.dbg) && wg < 1024 && threadIdx.x == 0 && blockIdx.z == 0) ? a.dbg + (size_t)wg * 64 : nullptr;
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }
    const float *qb = a.q + (int64_t)i * dim + (int64_t)kvh * r2 * hs;
    float *sb       = a.scores + ((int64_t)i * a.n_heads + (int64_t)kvh * r2) * a.n_ctx;
    float qf[R2MAX][NV];
#pragma unroll
    for (int g = 0; g < R2MAX; g++)
#pragma unroll
        for (int m = 0; m < NV; m++) qf[g][m] = (g < r2) ? qb[g * hs + m * 32 + c] : 0.f;
    float kf[4][NV];
#pragma unroll
    for (int rd = 0; rd < 4; rd++) {
        const int j     = j0 + rd * 8 + hw;
        const float *kr = a.k_cache + (int64_t)(j < n_kv ? j : 0) * kvd + kvh * hs;
#pragma unroll
        for (int m = 0; m < NV; m++) kf[rd][m] = kr[m * 32 + c];
    }
    if (dbg) dbg[1] = __builtin_amdgcn_s_memtime(); // loads issued
#pragma unroll
    for (int rd = 0; rd < 4; rd++) {
        const int j     = j0 + rd * 8 + hw;
        const bool live = j < n_kv;
#pragma unroll
        for (int g = 0; g < R2MAX; g++) {
            if (g < r2) {
                float s = 0.f;
#pragma unroll
                for (int m = 0; m < NV; m++) s = __fmaf_rn(kf[rd][m], qf[g][m], s); // sum = x*y + sum, x = K row (src0)
                s = reduce_f32x8x4(s);
                if (live && c == 0) sb[(int64_t)g * a.n_ctx + j] = s;
                if (dbg && rd == 0 && g == 0 && s != 12345.678f) dbg[2] = __builtin_amdgcn_s_memtime(); // q and the first K round have landed
            }
        }
    }
    if (dbg) { dbg[3] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

// Batches: the same scores on the matrix cores, still bit-exact.  v_mfma_f32_16x16x4_f32 is a k-ordered f32 fma chain
// (D = fma(a_k3, b_k3, fma(a_k2, b_k2, fma(a_k1, b_k1, fma(a_k0, b_k0, C)))), one rounding per product), and chain c of
// ggml_vec_dot_f32 is exactly such a chain over the elements c, c + 32, c + 64, c + 96 of the head: one MFMA per chain c
// with k = the 32-element block index gives the 16 x 16 tile of chain-c partials (16 positions x 16 (column, head)
// pairs), and GGML_F32x8_REDUCE becomes 31 element-wise adds of tiles in its own association.  No cross-lane reduction
// at all; a K row is fetched once per 16 positions and meets every column of the batch.
// grid (ceil(n_ctx / 64), n_kv_heads, SCM_Z): one wave per 16 positions, the column tiles split over blockIdx.z.
typedef float ps_f32x4 __attribute__((ext_vector_type(4)));
constexpr int SCM_Z = 8;
// Operands go through LDS: a lane's operand run is 128 B of one row, and fetched directly a wave instruction touches 64
// cache lines for 1 KiB (the CU's address unit, not the matrix core, set the time).  So a wave fetches its 16 K rows with
// row-coalesced instructions into its own LDS block once, the workgroup fetches each 16-column block of q once for its
// four waves (double-buffered, one barrier per column tile), and the operand registers are read back from rows padded by
// 16 B (conflict-free b128 reads).
template <int NV>
__global__ __launch_bounds__(256, 2) void attn_scores_mfma_kernel(psl_attn_args a) {
    // LDS rows of hs floats, UNPADDED, the 16-byte column index XOR-ed with (row & 7): the matrix operand's lanes (row = lane & 15, k-group = lane >> 4)
    // are served by ds_read_b128 in the lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md), for which no padded row stride is free of
    // conflicts (rows of hs + 4 floats: two passes on every operand read, 33 % of the kernel's LDS cycles in profiles/r03_pmc_sq_prefill.txt);
    // the swizzle is (tools/lds_bank_check.py: one pass, reads and parks, head sizes 64 and 128)
    constexpr int hs = NV * 32, RS = hs, SEGS = hs / 4, NSK = 16 * SEGS / 64, NSQ = (16 * SEGS + 255) / 256; // float4 segments per lane: K block / q block
    auto swz = [](int row, int seg) { return row * RS + ((seg ^ (row & 7)) << 2); }; // float index of 16-byte column `seg` of row `row`
    const int dim = a.n_heads * hs, kvd = a.n_kv_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kvh = blockIdx.y, bs = att_bs(a), n_kv = att_pos0(a) + bs;
    if ((int)blockIdx.x * 64 >= n_kv) return; // (the whole workgroup)
    const int j0 = ((int)blockIdx.x * 4 + wave) * 16;
    const bool live = j0 < n_kv; // (a wave past the end still helps fetching q and keeps the barriers)
    __shared__ __attribute__((aligned(16))) float kl[4][16 * RS];
    __shared__ __attribute__((aligned(16))) float ql[2][16 * RS];
    const int rl = lane & 15, m = lane >> 4; // A: row rl (position), k = m;  B: k = m, column rl
    const int N = bs * r2, n_tiles = (N + 15) / 16;
    float4 sq0, sq1; // (named, not an array: the array went to scratch)
    auto q_fetch1 = [&](int tile, int k) -> float4 { // (segment indices past the block are clamped here and not parked)
        const int sidx = min((int)threadIdx.x + k * 256, 16 * SEGS - 1), row = sidx / SEGS, sg = sidx - row * SEGS;
        const int n = min(tile * 16 + row, N - 1), i = n / r2, g = n - i * r2;
        return *(const float4 *)(a.q + (int64_t)i * dim + (int64_t)(kvh * r2 + g) * hs + sg * 4);
    };
    auto q_fetch = [&](int tile) { sq0 = q_fetch1(tile, 0); if (NSQ > 1) sq1 = q_fetch1(tile, 1); };
    auto q_park1 = [&](float *buf, int k, const float4 v) {
        const int sidx = (int)threadIdx.x + k * 256, row = sidx / SEGS, sg = sidx - row * SEGS;
        if (sidx < 16 * SEGS) *(float4 *)(buf + swz(row, sg)) = v;
    };
    auto q_park = [&](float *buf) { q_park1(buf, 0, sq0); if (NSQ > 1) q_park1(buf, 1, sq1); };
    static_assert(NSQ <= 2, "q block: at most two segments per thread");
    // ---- this wave's K block, then the first q block
    {
        float4 sk[NSK];
#pragma unroll
        for (int k = 0; k < NSK; k++) {
            const int sidx = lane + k * 64, row = sidx / SEGS, sg = sidx - row * SEGS;
            sk[k] = *(const float4 *)(a.k_cache + (int64_t)min(j0 + row, n_kv - 1) * kvd + kvh * hs + sg * 4);
        }
        q_fetch((int)blockIdx.z < n_tiles ? (int)blockIdx.z : 0);
#pragma unroll
        for (int k = 0; k < NSK; k++) {
            const int sidx = lane + k * 64, row = sidx / SEGS, sg = sidx - row * SEGS;
            *(float4 *)(&kl[wave][swz(row, sg)]) = sk[k];
        }
        q_park(ql[0]);
    }
    __syncthreads();
    // A operand of chain c: K[j0 + rl][32 m + c]
    float ka[32];
#pragma unroll
    for (int q4 = 0; q4 < 8; q4++) {
        const float4 t = (m < NV) ? *(const float4 *)(&kl[wave][swz(rl, 8 * (m < NV ? m : 0) + q4)]) : make_float4(0.f, 0.f, 0.f, 0.f);
        ka[4 * q4] = t.x; ka[4 * q4 + 1] = t.y; ka[4 * q4 + 2] = t.z; ka[4 * q4 + 3] = t.w;
    }
    const ps_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int tile = blockIdx.z; tile < n_tiles; tile += SCM_Z, cur ^= 1) {
        const bool more = tile + SCM_Z < n_tiles;
        q_fetch(more ? tile + SCM_Z : tile); // (the next block: one column tile of matrix work to hide behind)
        if (live) {
            const int n = min(tile * 16 + rl, N - 1), i = n / r2, g = n - i * r2;
            float qb[32]; // B operand of chain c: q[i][kvh * r2 + g][32 m + c]
#pragma unroll
            for (int q4 = 0; q4 < 8; q4++) {
                const float4 t = (m < NV) ? *(const float4 *)(&ql[cur][swz(rl, 8 * (m < NV ? m : 0) + q4)]) : make_float4(0.f, 0.f, 0.f, 0.f);
                qb[4 * q4] = t.x; qb[4 * q4 + 1] = t.y; qb[4 * q4 + 2] = t.z; qb[4 * q4 + 3] = t.w;
            }
            auto chain = [&](int c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(ka[c], qb[c], zero, 0, 0, 0); }; // sum = x*y + sum, x = K row
            // GGML_F32x8_REDUCE: (c, c+16), then (.., c+8), then (.., c+4); the two hadds below.  Eight products at a time
            // (all 32 at once cost 128 accumulator registers), and the next eight are issued BEFORE the adds of the
            // current eight: the matrix core works while the vector unit folds.
            ps_f32x4 t3[4], pr[2][8];
            auto issue = [&](ps_f32x4 (&d)[8], int c) {
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = chain(c + 4 * k); // c, c+4, ..., c+28
            };
            auto fold = [&](const ps_f32x4 (&d)[8]) { // d[k] = chain(c + 4k)
                const ps_f32x4 ta = d[0] + d[4], tb = d[2] + d[6]; // (c, c+16), (c+8, c+24)
                const ps_f32x4 tc = d[1] + d[5], td = d[3] + d[7]; // (c+4, c+20), (c+12, c+28)
                return (ta + tb) + (tc + td);
            };
            // (an empty asm ties the B operands of group g to the folded sum of group g - 2: the products are pure values with no
            // chain edge, and without the tie all 32 are issued up front -- 128 live accumulator registers -- whatever the barriers say)
            auto tie = [&](int g, const ps_f32x4 &t) {
                asm volatile("" : "+v"(qb[g]), "+v"(qb[g + 4]), "+v"(qb[g + 8]), "+v"(qb[g + 12]), "+v"(qb[g + 16]), "+v"(qb[g + 20]), "+v"(qb[g + 24]), "+v"(qb[g + 28])
                             : "v"(t[0]), "v"(t[2]));
            };
            issue(pr[0], 0);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c < 3) {
                    if (c >= 1) tie(c + 1, t3[c - 1]);
                    issue(pr[(c + 1) & 1], c + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                t3[c] = fold(pr[c & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            const ps_f32x4 res = (t3[0] + t3[1]) + (t3[2] + t3[3]);
            // D: column = lane & 15 (the (i, g) pair), rows 4 * (lane >> 4) + r (positions)
            if (tile * 16 + rl < N) {
                float *sb = a.scores + ((int64_t)i * a.n_heads + (int64_t)kvh * r2 + g) * a.n_ctx;
                const int jb = j0 + 4 * m;
                if (jb + 3 < n_kv) *(float4 *)(sb + jb) = make_float4(res[0], res[1], res[2], res[3]);
                else
#pragma unroll
                    for (int r = 0; r < 4; r++) if (jb + r < n_kv) sb[jb + r] = res[r];
            }
        }
        q_park(ql[cur ^ 1]);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- softmax + V·p in one launch
// out[i][h][d] = V^T[kvh*hs + d][0..n_kv) · softmax(scale·s[i][h][:] + mask)
// grid (hs/4, n_kv_heads, bs), 256 threads.  Phase 1: one wave per q head of the kv group turns its raw score row
// into probabilities held in LDS (scale + mask, max, ggml_v_expf on groups of 8 with the in-group sum tree, libm
// expf on the n_kv % 8 tail, double row sum, p = e * (float)(1/sum) — ggml.c:14889-14925, :2814-2863).  Phase 2: the
// four V rows of this workgroup are streamed through LDS in tiles with every thread's loads in flight at once, and
// 8 half-waves run the 32-lane fp32 chains of ggml_vec_dot_f32 in position order (chain state stays in registers
// across tiles), then GGML_F32x8_REDUCE and the n_kv % 32 leftovers.
constexpr int PV_TILE = 2560;          // V columns per LDS tile (4 channels x PV_TILE floats)
constexpr int PV_VSTR = PV_TILE + 32;  // row stride: the two half-waves of a wave read different channels -> disjoint banks
constexpr int PV_NT   = 1024;          // one workgroup per CU
constexpr int PV_NW   = PV_NT / 64;
constexpr int PV_LPT  = (4 * PV_TILE / 4 + PV_NT - 1) / PV_NT; // float4 V loads per thread and tile
// Exact softmax numerators of the r2 rows of (kv head kvh, column i) into LDS (pl[g][n_kv4]) and 1/sum per row (invs):
// ggml_vec_soft_max_f32 (ggml.c:2831-2866).  All PV_NT threads; ends with the rows complete and invs visible after the
// caller's next __syncthreads().
__device__ __forceinline__ void attn_softmax_rows(const psl_attn_args &a, const int i, const int kvh, const int r2, const int pos0, const int bs,
                                                  const int n_kv, float *pl, float (*redf)[PV_NW], double (*redd)[PV_NW], float *invs,
                                                  unsigned long long *dbg = nullptr) {
    auto mark = [&](int k) { if (dbg) dbg[k] = __builtin_amdgcn_s_memtime(); };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n8 = n_kv & ~7, n_kv4 = (n_kv + 3) & ~3;
    // ---- phase 1a: scores -> masked, scaled logits in LDS; row maxima
    constexpr int EPT = 3; // elements per thread, row and trip: one trip covers n_kv <= 3072
    float rmax[R2MAX];
#pragma unroll
    for (int g = 0; g < R2MAX; g++) rmax[g] = -INFINITY;
    for (int j0 = threadIdx.x; j0 < n_kv; j0 += PV_NT * EPT) {
        float sv[R2MAX][EPT];
#pragma unroll
        for (int g = 0; g < R2MAX; g++)
#pragma unroll
            for (int t = 0; t < EPT; t++) {
                const int j = j0 + PV_NT * t;
                const float *sp = a.scores + ((int64_t)i * a.n_heads + (int64_t)kvh * r2 + g) * a.n_ctx + j;
                sv[g][t] = (g < r2 && j < n_kv) ? *sp : 0.f;
            }
#pragma unroll
        for (int t = 0; t < EPT; t++) {
            const int j = j0 + PV_NT * t;
            if (j < n_kv) {
                const bool ok = (j < pos0) ? (a.kv_vis ? a.kv_vis[j] != 0 : true) : (a.tree ? a.tree[i * bs + (j - pos0)] != 0 : (j - pos0) <= i);
#pragma unroll
                for (int g = 0; g < R2MAX; g++) {
                    if (g < r2) {
                        float v = __fmul_rn(sv[g][t], a.scale);
                        v       = __fadd_rn(v, ok ? 0.f : -INFINITY);
                        pl[(size_t)g * n_kv4 + j] = v;
                        rmax[g] = fmaxf(rmax[g], v);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < R2MAX; g++) {
        if (g < r2) { const float m = wave_max_dpp(rmax[g]); if (lane == 0) redf[g][wave] = m; }
    }
    mark(2); // scores landed, logits in LDS
    __syncthreads();
    mark(3);
    // ---- phase 1b: e_j = exp(x_j - max), row sums.  r2 | 16: wave w works on row w % r2 with 16/r2 - 1 others,
    //      else one wave per row
    const bool split = (PV_NW % r2) == 0;
    const int wpr    = split ? PV_NW / r2 : 1;
    {
        const int g = split ? wave % r2 : wave, sub = split ? wave / r2 : 0;
        if (g < r2) {
            const int slot = sub * 64 + lane, nthr = wpr * 64;
            float mx = redf[g][0];
#pragma unroll
            for (int w = 1; w < PV_NW; w++) mx = fmaxf(mx, redf[g][w]);
            float *pg = pl + (size_t)g * n_kv4;
            double rs = 0.0;
            for (int gi = slot; gi * 8 < n8; gi += nthr) { // one lane per group of 8: the reference's in-group association
                const float4 lo = *(const float4 *)(pg + gi * 8), hi = *(const float4 *)(pg + gi * 8 + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int l = 0; l < 8; l++) v[l] = ps_v_expf(__fsub_rn(v[l], mx)); // (ps_v_expf_n here costs attn_softmax_pv_cols_kernel<2> five spilled registers)
                *(float4 *)(pg + gi * 8)     = make_float4(v[0], v[1], v[2], v[3]);
                *(float4 *)(pg + gi * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                const float a0 = __fadd_rn(v[4], v[0]), a1 = __fadd_rn(v[5], v[1]), a2 = __fadd_rn(v[6], v[2]), a3 = __fadd_rn(v[7], v[3]);
                rs += (double)__fadd_rn(__fadd_rn(a0, a2), __fadd_rn(a1, a3));
            }
            if (slot == 0)
                for (int j = n8; j < n_kv; j++) { const float e = ps_expf_glibc(__fsub_rn(pg[j], mx)); pg[j] = e; rs += (double)e; }
            const double sw = wave_sum_d_dpp(rs);
            if (lane == 0) redd[g][sub] = sw;
        }
    }
    mark(4);
    __syncthreads();
    mark(5);
    if ((int)threadIdx.x < r2) {
        double t = 0.0;
        for (int w = 0; w < wpr; w++) t += redd[threadIdx.x][w];
        invs[threadIdx.x] = (float)(1.0 / t);
    }
}

// Fused softmax + V·p for one (4-channel group, kv head, batch column): grid (hs/4, n_kv_heads, bs), 1024 threads.
//   t = 0   the first V tile and every score of the r2 rows are requested together (independent streams)
//   phase 1 exact softmax numerators e_j (ggml_vec_soft_max_f32, ggml.c:2831-2866: ggml_v_expf on groups of 8 with
//           the in-group sum tree, libm expf on the n%8 tail, double sum), rows spread over the 16 waves
//   phase 2 out[h][d] = ggml_vec_dot_f32(V[d], p[h]) (ggml.c:2126-2160): 32 fma chains over columns 32i+c, one lane
//           each, p_j = e_j * (float)(1/sum) formed at the read (same rounding as the reference's scale pass),
//           GGML_F32x8_REDUCE tree, n%32 leftovers (mul, add).  V comes through LDS tiles (prefetched one ahead).
__global__ __launch_bounds__(PV_NT) void attn_softmax_pv_kernel(psl_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float pl[]; // [r2][n_ctx4] e_j, then [4][PV_VSTR] V tile
    const int hs = a.head_size, dim = a.n_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int kvh = blockIdx.y, i = blockIdx.z, bs = att_bs(a), pos0 = att_pos0(a);
    const int n_kv = pos0 + bs, np = n_kv & ~31, n_kv4 = (n_kv + 3) & ~3;
    float *vt = pl + (size_t)r2 * (((size_t)a.n_ctx + 3) & ~(size_t)3);
    __shared__ float redf[R2MAX][PV_NW];
    __shared__ double redd[R2MAX][PV_NW];
    __shared__ float invs[R2MAX];
    const int wg = blockIdx.y * gridDim.x + blockIdx.x; // timeline (tools/gpu_attn_timeline.py, key 41)
    unsigned long long *const dbg = (PS_TL(a.dbg) && wg < 1024 && threadIdx.x == 0 && blockIdx.z == 0) ? a.dbg + (size_t)wg * 64 : nullptr;
    auto mark = [&](int k) { if (dbg) dbg[k] = __builtin_amdgcn_s_memtime(); };
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }

    // ---- V tile loads (registers -> LDS), one tile ahead of the chains
    const float *vbase = a.v_cache + ((int64_t)kvh * hs + blockIdx.x * 4) * a.n_ctx;
    float4 ld[PV_LPT];
    auto load_tile = [&](int t0) { // columns [t0, t0 + PV_TILE) of the 4 channel rows (zero past n_kv)
#pragma unroll
        for (int k = 0; k < PV_LPT; k++) {
            const int f = threadIdx.x + PV_NT * k, row = f / (PV_TILE / 4), col = (f % (PV_TILE / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < 4 && t0 + col < n_kv4) v = *(const float4 *)(vbase + (int64_t)row * a.n_ctx + t0 + col); // n_kv4 <= n_ctx4: in bounds
            ld[k] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < PV_LPT; k++) {
            const int f = threadIdx.x + PV_NT * k, row = f / (PV_TILE / 4), col = (f % (PV_TILE / 4)) * 4;
            if (row < 4) *(float4 *)(vt + row * PV_VSTR + col) = ld[k];
        }
    };
    load_tile(0);
    mark(1);
    attn_softmax_rows(a, i, kvh, r2, pos0, bs, n_kv, pl, redf, redd, invs, dbg);
    mark(6);
    store_tile();
    mark(7);
    __syncthreads();
    mark(8);

    // ---- phase 2: V·p.  thread: chain c = tid & 31, channel dl = (tid >> 5) & 3, head g = tid >> 7
    const int c = threadIdx.x & 31, dl = (threadIdx.x >> 5) & 3, g = threadIdx.x >> 7;
    const bool live = g < r2;
    const float inv = live ? invs[g] : 0.f;
    const float *pg = pl + (size_t)(live ? g : 0) * n_kv4;
    const float *vr = vt + dl * PV_VSTR;
    float acc = 0.f, out = 0.f;
    for (int t0 = 0; t0 < n_kv; t0 += PV_TILE) {
        const int tn = min(PV_TILE, np - t0); // chain columns of this tile (multiple of 32, may be <= 0 on a leftover-only tile)
        if (t0 > 0) {
            __syncthreads(); // previous tile consumed
            store_tile();
            __syncthreads();
        }
        if (t0 + PV_TILE < n_kv) load_tile(t0 + PV_TILE); // next tile streams in while this one is consumed
        if (live) {
            int j = c;
            for (; j + 7 * 32 < tn; j += 8 * 32) { // eight LDS round trips in flight per chain step group
                float v[8], e[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { v[k] = vr[j + 32 * k]; e[k] = pg[t0 + j + 32 * k]; }
#pragma unroll
                for (int k = 0; k < 8; k++) acc = __fmaf_rn(v[k], __fmul_rn(e[k], inv), acc); // x = V row (src0), y = p
            }
            for (; j < tn; j += 32) acc = __fmaf_rn(vr[j], __fmul_rn(pg[t0 + j], inv), acc);
            if (t0 + PV_TILE >= n_kv) { // last tile: reduce the 32 chains, then the n%32 leftovers in order
                float sres = reduce_f32x8x4(acc);
                for (int jj = np; jj < n_kv; jj++) sres = ps_dot_left(sres, vr[jj - t0], __fmul_rn(pg[jj], inv), jj - np, n_kv - np);
                out = sres;
            }
        }
    }
    mark(9);
    if (live && c == 0) a.att[(int64_t)i * dim + ((int64_t)kvh * r2 + g) * hs + blockIdx.x * 4 + dl] = out;
    if (dbg) { dbg[10] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

// ---------------------------------------------------------------- single token: scores + soft-max + V.p in ONE launch
// attn_decode2_kernel: K·q, soft-max and V·p of one cached token in one launch whose critical path is
//   q / K landed -> scores -> ONE exchange -> max -> exp -> sum -> 17 dependent matrix instructions -> reduce.
// What the round-2 timeline of the two launches showed to be waste is gone (profiles/r02_attention_timeline.txt):
//   * the K rows below the host's lower bound of the cache length (n_kv_lo: a HINT, results never depend on it) are
//     requested before the device-resident position has arrived (that load is an L2 miss on every launch); the loads in
//     front of that wait are unconditional so that it is an exact count; q is fetched once per workgroup (through LDS), not
//     once per half-wave: the start-up was bound by the ISSUE of 50 dword loads per lane.  With 1024 threads every
//     instruction costs 16 wave issues per CU (the first version spent 1.2 us ISSUING the scores and 2 us on the scale /
//     mask / address arithmetic around the gather): eight lanes per cached position hold a K row as four 16-byte loads, a
//     lane runs four of ggml_vec_dot_f32's chains over all their steps and GGML_F32x8_REDUCE is three row shifts on four
//     values plus three in-lane adds — 15 instead of 56 instructions per eight positions and head, and a wave whose slice
//     lies behind the cache length does nothing at all;
//   * the scores are exchanged once per kv head: write-through (sc1) stores, drained, one ticket per workgroup on the kv
//     head's counter (epoch = ticket / workgroups per head: nothing is ever reset), ONE lane polls that ONE word, then every
//     lane fetches its scores straight into the soft-max's register layout (lane = head, group of 8 positions — the
//     reference's vector loop, ggml.c:2831-2866).  The V rows are requested BEHIND the drain (in front of it the drain
//     would wait for them, too).  (Tried first: scores as self-describing {tag, value} granules polled by every lane —
//     correct: profiles/r03_micro_stale.txt shows that an agent-scope load never returns a line from before another
//     workgroup's write-through store — but 256 x 1024 polling lanes starve the very stores they wait for: 40 us per
//     exchange, and 30 ms time-outs in one run.)
//   * the n_kv % 8 leftovers take libm's expf on lanes of their own, in parallel; e_j goes to LDS as it is and
//     p_j = e_j * (float)(1/sum) is formed where it is used (the same single rounding): one barrier less;
//   * V·p runs on the matrix cores: v_mfma_f32_16x16x4_f32 is a k-ordered fma chain with one rounding per product
//     (attn_pv_mfma_kernel above).  A workgroup owns 4 channels x r2 <= 4 heads, so one instruction carries FOUR of
//     ggml_vec_dot_f32's 32 chains: A row = (chain ci, channel), B column = (chain cj, head), k = four successive
//     32-position blocks, and the 4 x 4 blocks with ci == cj of the 16 x 16 result are the chains' partial sums (the
//     others are discarded).  Eight waves, one per group of four chains, 17 dependent instructions at n_kv = 2100
//     instead of 66 LDS round trips per lane.
// grid: (head_size / 4) x n_kv_heads workgroups of 1024 threads, linear id = x * n_kv_heads + kv head (with 8 kv heads a
// head's workgroups share an XCD, i.e. an L2); every workgroup must be resident (the host checks the grid against the CU
// count; the poll is bounded and raises sync[31]).
// LDS: e [4][RS] and V [4][RS] (RS = n_ctx rounded up to 128, + 36: rows 4 banks apart), n_ctx <= 4096.
// NT threads: 1024, or 512 (tried for its shorter kernel boundary; the per-lane work doubles and the launch measured 0.4 us slower).  LPH = NT / 4 lanes per head take groups of 8 positions: LPH x 8 x TRIPS >= n_ctx
constexpr int D2_MAXCTX = 4096;
// (round 6) KVS = psl_attn_args::kv_stream: the cached K rows and V channels of a single-token step are read ONCE per token; when the model's whole cache is larger than the
// memory-side cache nothing of it survives to the next token, and read with plain loads it displaces everything else there and in the L2s: non-temporal loads then
// (8B at n_kv 2048, 537 MB of cache: +2.1 % decode in the fused launch, +0.8 % here; profiles/r06_kv_nt_ab.txt, r06_nt_more_ab.txt).  A small cache (Llama-3.2-1B at 640
// positions: 42 MB) IS served from there token after token: plain loads (1548 against 1517 tok/s with the hint, profiles/r06_ad2_nt_small_models.txt).
template <int KVS>
__device__ __forceinline__ float4 ad2_ld4(const float *p) {
    if (!KVS) return *(const float4 *)p;
    const ps_u32x4 t = __builtin_nontemporal_load((const ps_u32x4 *)p);
    float4 r;
    __builtin_memcpy(&r, &t, 16);
    return r;
}
template <int NV, int NT, int KVS>
__global__ __launch_bounds__(NT) void attn_decode2_kernel(psl_attn_args a) {
    constexpr int NW = NT / 64, SPP = NW / 4, LPH = NT / 4, WPH = LPH / 64, D2_TRIPS = D2_MAXCTX / (8 * LPH), VCH = D2_MAXCTX / (4 * NT); // SPP: slices per pass
    extern __shared__ __attribute__((aligned(16))) float d2s[];
    constexpr int hs = NV * 32, RMAX = 16 / NV, G = hs / 4; // G workgroups per kv head; slice (32 positions) s belongs to workgroup s % G
    const int kvd = a.n_kv_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int kvh = (int)blockIdx.x % a.n_kv_heads, bx = (int)blockIdx.x / a.n_kv_heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RS = ((a.n_ctx + 127) & ~127) + 36, XS = (a.n_ctx + 31) & ~31; // (XS: a 128-byte line of the exchange buffer belongs to one workgroup)
    float *const pl = d2s, *const vt = pl + 4 * RS, *const red = vt + 4 * RS, *const pleft = red + 512, *const redf = pleft + 128;
    float *const tails = redf + 16, *const qs = tails + 32, *const sst = qs + 4 * hs; // sst [RMAX][4][32]: scores staged for whole-row stores
    double *const redd = (double *)(sst + RMAX * 128);
    uint64_t *const etab = (uint64_t *)(redd + 16); // glibc's expf table (a constant-memory lookup behind the V requests costs > 1 us)
    unsigned long long *const dbg = (PS_TL(a.dbg) && blockIdx.x < 1024 && tid == 0) ? a.dbg + (size_t)blockIdx.x * 64 : nullptr; // timeline key 42
    auto mark = [&](int k) { if (dbg) dbg[k] = __builtin_amdgcn_s_memtime(); };
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }

    // ---- t = 0.  The position first (a vector load on purpose: the scalar counter is shared with the kernel arguments and
    //      LDS), then q (once per workgroup, through LDS) and the K rows the host knows to exist — all UNCONDITIONAL (clamped
    //      addresses), so that the wait for the position is an exact count and does not wait for the K rows behind it
    const ps_step_state *sp = a.state;
    asm volatile("" : "+v"(sp)); // (opaque: keeps it a vector load, issued here; global address space: a flat load would tie up both counters)
    const int st_pos0 = *(const __attribute__((address_space(1))) int *)(uintptr_t)sp; // (a dword: unused lanes of a wider load get reused as temporaries, which waits for the load)
    const int uw = __builtin_amdgcn_readfirstlane(wave); // (provably uniform: scalar branches)
    const int p8 = lane >> 3, tq = lane & 7;              // eight lanes per cached position; lane tq owns chains 4 tq .. 4 tq + 3
    constexpr int PASSES = RMAX / SPP;                     // a pass = NW waves x 8 positions = SPP 32-position slices
    const int nq4 = r2 * hs / 4; // float4 units of this kv head's q rows
    const float4 q4 = *(const float4 *)(a.q + (int64_t)kvh * r2 * hs + (tid < nq4 ? tid : 0) * 4);
    const uint64_t et_w = ps_exp2f_tab[tid & (PS_EXP2F_N - 1)];
    const float *kb = a.k_cache + kvh * hs + 4 * tq;
    float4 kf[PASSES][NV]; // K row of this lane's position: elements 32 i + 4 tq .. + 3 = step i of chains 4 tq .. 4 tq + 3
    const int nlo = a.n_kv_lo;
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int sl = bx + ((uw >> 2) + SPP * ps) * G;
        const bool early = sl * 32 + 32 <= nlo && sl * 32 + 32 <= a.n_ctx; // (uniform)
        const float *kr = kb + (int64_t)(early ? sl * 32 + (uw & 3) * 8 + p8 : 0) * kvd; // (not hinted: row 0 once more, a cache hit)
#pragma unroll
        for (int m = 0; m < NV; m++) kf[ps][m] = ad2_ld4<KVS>(kr + m * 32);
    }
    mark(1);
    const int pos0 = __builtin_amdgcn_readfirstlane(st_pos0); // (uniform: everything derived from it is scalar control flow)
    const int n_kv = pos0 + 1, n8 = n_kv & ~7, np = n_kv & ~31, nblk = np >> 5, n_it = (nblk + 3) >> 2, np_pad = n_it * 128;
    const int ntail = n_kv - n8, nleft = n_kv - np;
    // ---- the rest of this workgroup's K rows (only the waves that have any: uniform branches)
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int sl = bx + ((uw >> 2) + SPP * ps) * G, j = sl * 32 + (uw & 3) * 8 + p8;
        const bool early = sl * 32 + 32 <= nlo && sl * 32 + 32 <= a.n_ctx;
        if (!early && sl * 32 + (uw & 3) * 8 < n_kv) {
            const float *kr = kb + (int64_t)(j < n_kv ? j : 0) * kvd;
#pragma unroll
            for (int m = 0; m < NV; m++) kf[ps][m] = ad2_ld4<KVS>(kr + m * 32);
        }
    }
    // ---- this workgroup's four V rows, columns below the cache length (registers; parked in LDS with the drain below).  Here,
    //      right behind K, so that the two streams are over before the exchange: the CU's vector-memory queue is first in, first
    //      out across its waves — requested later, the V rows sit in front of the polls, of the gathered scores or of whatever
    //      runs next (measured: the 8.6 MB cost 1.4 us wherever they were put).  Load k = row k, columns 4 tid .. 4 tid + 3
    const float *vbase = a.v_cache + ((int64_t)kvh * hs + bx * 4) * a.n_ctx;
    float4 vld[VCH][4];
#pragma unroll
    for (int cc = 0; cc < VCH; cc++) {
        const int col = 4 * (tid + NT * cc);
        const bool v_in = col < ((n_kv + 3) & ~3); // (n_kv rounded up to 4 <= n_ctx: in bounds)
#pragma unroll
        for (int k = 0; k < 4; k++) vld[cc][k] = ad2_ld4<KVS>(vbase + (v_in ? (int64_t)k * a.n_ctx + col : 0));
    }
    if (tid < nq4) *(float4 *)(qs + tid * 4) = q4;
    if (tid < PS_EXP2F_N) etab[tid] = et_w;
    mark(2);
    __syncthreads();

    // ---- scores of this workgroup's positions, stored write-through.  ggml_vec_dot_f32 (ggml.c:2092-2133): chain c = 8 a + u runs
    //      over the elements 32 i + c; here lane tq holds chains 4 tq + e (e = 0..3) with all their steps i.  GGML_F32x8_REDUCE
    //      (ggml.c:1354-1371): acc0 += acc2, acc1 += acc3 = chain c + chain c + 16 = lane tq + lane tq + 4; acc0 += acc1 = lane + 2;
    //      low + high 128 bits = lane + 1; then the two hadds inside lane 0: (x0 + x1) + (x2 + x3)
    float *const xb = a.xchg + (size_t)kvh * 4 * XS;
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) {
        const int sl = bx + ((uw >> 2) + SPP * ps) * G;
        if (sl * 32 + (uw & 3) * 8 < n_kv) { // (uniform: a wave behind the cache length does nothing)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (g < r2) { // (uniform)
                    float x[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < NV; m++) {
                        const float4 qv = *(const float4 *)(qs + g * hs + m * 32 + 4 * tq);
                        x[0] = __fmaf_rn(kf[ps][m].x, qv.x, x[0]); // sum = x*y + sum, x = K row (src0)
                        x[1] = __fmaf_rn(kf[ps][m].y, qv.y, x[1]);
                        x[2] = __fmaf_rn(kf[ps][m].z, qv.z, x[2]);
                        x[3] = __fmaf_rn(kf[ps][m].w, qv.w, x[3]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) x[e] = __fadd_rn(x[e], dpp_f<0x104>(x[e])); // tq < 4: + lane tq + 4
#pragma unroll
                    for (int e = 0; e < 4; e++) x[e] = __fadd_rn(x[e], dpp_f<0x102>(x[e])); // tq < 2: + lane tq + 2
#pragma unroll
                    for (int e = 0; e < 4; e++) x[e] = __fadd_rn(x[e], dpp_f<0x101>(x[e])); // tq = 0: + lane 1
                    const float s = __fadd_rn(__fadd_rn(x[0], x[1]), __fadd_rn(x[2], x[3]));
                    if (tq == 0) sst[(((uw >> 2) + SPP * ps) * 4 + g) * 32 + (uw & 3) * 8 + p8] = s;
                }
            }
        }
    }
    // one 128-byte row per (slice, head), a whole line per store instruction: 8 positions x 4 bytes at a time made every line a
    // read-modify-write at the memory side (drain 0.5 -> 0.3 us, the counter complete 0.9 us earlier)
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RMAX * 4 / NW; rr++) {
        const int row = uw + NW * rr, rd = row >> 2, gg = row & 3, sl = bx + rd * G;
        if (sl * 32 < n_kv && gg < r2 && lane < 32) coh_store_f(xb + (size_t)gg * XS + sl * 32 + lane, sst[row * 32 + lane]);
    }
    mark(3);
    // ---- exchange: every wave drains its stores (and its V rows), one ticket per workgroup, ONE lane polls the kv head's counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // V rows into LDS (they landed with the drain); zeros past the cache length: the matrix instructions run over whole groups of
    // 4 blocks of positions
#pragma unroll
    for (int cc = 0; cc < VCH; cc++) {
        const int col = 4 * (tid + NT * cc);
        if (col < RS - 36) {
            const bool v_in = col < ((n_kv + 3) & ~3);
#pragma unroll
            for (int k = 0; k < 4; k++) *(float4 *)(vt + k * RS + col) = v_in ? vld[cc][k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    mark(4);
    if (wave == 0) { // (uniform) the polling wave asks for its V rows AFTER the poll: loads return in order
        if (tid == 0) {
            unsigned *ctr = a.tick + kvh * 64; // (a 256-byte stretch per kv head)
            const unsigned ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (ticket / (unsigned)G + 1u) * (unsigned)G; // exactly G arrivals per head and launch
            int spins = 0;
            while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 16)) { __hip_atomic_store(a.sync + 31, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; } // never hang the GPU: the host turns the flag into an error
            }
        }
        asm volatile("" ::: "memory"); // (nothing below moves above the poll)
    }
    __syncthreads();
    mark(5);

    // ---- gather: lane (head g, t) takes the groups of 8 positions t, t + LPH, ...  Plain loads: this launch has not touched
    //      these lines before the counter completed (L1 and L2 hold nothing older than the kernel boundary), every byte was
    //      written through and acknowledged before its workgroup's ticket, and the 32 workgroups of a kv head share an L2
    const int g = tid / LPH, t = tid % LPH;
    const bool hl = g < r2;
    const float *xg = xb + (size_t)(hl ? g : 0) * XS;
    float sv[D2_TRIPS][8];
#pragma unroll
    for (int k = 0; k < D2_TRIPS; k++) {
        const int j0 = (t + LPH * k) * 8;
        const float *src = xg + ((hl && j0 < n_kv) ? j0 : 0); // (unconditional, clamped)
        const float4 lo = *(const float4 *)src, hi = *(const float4 *)(src + 4);
        sv[k][0] = lo.x; sv[k][1] = lo.y; sv[k][2] = lo.z; sv[k][3] = lo.w; sv[k][4] = hi.x; sv[k][5] = hi.y; sv[k][6] = hi.z; sv[k][7] = hi.w;
    }
    // ---- scale + mask (softmax_ext, ggml.c:14889-14914), row maxima.  (Without hidden slots the mask term is + 0.0f, which
    //      changes no value the soft-max can tell apart: skipped.)
    float lmax = -INFINITY;
    if (a.kv_vis) { // (uniform) hidden cache slots (KVCacheInterface::mask): -inf before the maximum
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
    {
        const float wm = wave_max_dpp(lmax);
        if (lane == 0) redf[wave] = wm; // waves WPH g .. WPH g + WPH - 1 belong to head g
    }
#pragma unroll
    for (int k = 0; k < D2_TRIPS; k++)
        if (hl && (t + LPH * k) * 8 == n8) { // the lane that holds the n_kv % 8 leftovers hands them to lanes of their own
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < ntail) tails[g * 8 + i] = sv[k][i];
        }
    __syncthreads();
    mark(6);
    float mx = redf[g * WPH];
#pragma unroll
    for (int w = 1; w < WPH; w++) mx = fmaxf(mx, redf[g * WPH + w]);
    // ---- e_j = exp(x_j - max): ggml_v_expf on whole groups of 8 with the reference's in-group sum tree, libm's expf on the
    //      n_kv % 8 leftovers (the last lanes of the head's last wave, one each), row sums in double (ggml.c:2831-2866).
    //      e goes to LDS as it is; p_j = e_j * (float)(1/sum) (ggml_vec_scale_f32) is formed where it is used, the same
    //      single rounding.  Whole blocks of 32 positions feed the chains (pl), positions past the last whole block are the
    //      dot product's leftovers (pleft), and the padding of the last group of 4 blocks is zero
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
    {
        const double sw = wave_sum_d_dpp(rs);
        if (lane == 0) redd[wave] = sw;
    }
    mark(7);
    __syncthreads();
    mark(8);

    // ---- V·p: wave w < 8 owns chains 4w .. 4w+3.  A: row rl = 4 ci + channel, k = block; B: k = block, column 4 cj + head
    if (wave < 8) { // (NT = 512: every wave)
        const int rl = lane & 15, kk = lane >> 4, ci = rl >> 2, rh = rl & 3;
        const bool bl = rh < r2;
        const int rb = bl ? rh : 0; // (heads past r2: head 0's row, masked below)
        double tot = redd[WPH * rb];
#pragma unroll
        for (int w = 1; w < WPH; w++) tot += redd[WPH * rb + w];
        const float inv = (float)(1.0 / tot);
        const float *va = vt + rh * RS + 4 * wave + ci + 32 * kk;
        const float *pb = pl + rb * RS + 4 * wave + ci + 32 * kk;
        ps_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // operands of four steps at a time, one group ahead of the matrix instructions (a dependent chain: 40 cycles each).
        // Steps past the last one (the group is rounded up) read the last step's V again against p = 0: fma(v, 0, acc) = acc
        float a0[4], b0[4], a1[4], b1[4];
        auto fetch = [&](float (&av)[4], float (&bv)[4], int s0) { // (reads only: scale and mask are applied where the operand is used)
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
                const float p = __fmul_rn(bv[i], inv); // p_j = e_j * (float)(1/sum)
                const float b = __uint_as_float(__float_as_uint(p) & ((bl && s0 + i < n_it) ? 0xffffffffu : 0u)); // (a mask, not a branch)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b, acc, 0, 0, 0); // sum = x*y + sum, x = V row (src0)
            }
        };
        if (n_it > 0) {
            fetch(a0, b0, 0);
            for (int s0 = 0; s0 < n_it; s0 += 8) {
                fetch(a1, b1, s0 + 4);
                __builtin_amdgcn_sched_barrier(0); // (the reads go out BEFORE the dependent matrix instructions stall the issue)
                chain(a0, b0, s0);
                if (s0 + 4 < n_it) { // (uniform)
                    fetch(a0, b0, s0 + 8);
                    __builtin_amdgcn_sched_barrier(0);
                    chain(a1, b1, s0 + 4);
                }
            }
        }
        // D: column = lane & 15 = 4 cj + head, rows 4 (lane >> 4) + r = 4 ci + channel r: the ci == cj blocks are the chains
        if ((lane >> 4) == ((lane & 15) >> 2)) {
#pragma unroll
            for (int r = 0; r < 4; r++) red[((4 * wave + (lane >> 4)) * 4 + r) * 4 + (lane & 3)] = acc[r];
        }
    }
    // the leftovers' operands meanwhile (the last 16 lanes of wave 15, which has no chains): v[jj], e[jj] * inv
    float lv[32], lp[32];
    if (tid >= NT - 16) {
        const int ch = (tid >> 2) & 3, h = tid & 3, hb = h < r2 ? h : 0;
        double tot = redd[WPH * hb];
#pragma unroll
        for (int w = 1; w < WPH; w++) tot += redd[WPH * hb + w];
        const float inv = (float)(1.0 / tot);
#pragma unroll
        for (int jj = 0; jj < 32; jj++) { // (np + 31 < RS: in bounds; entries past the cache length are never added)
            lv[jj] = vt[ch * RS + np + jj];
            lp[jj] = __fmul_rn(pleft[hb * 32 + jj], inv);
        }
    }
    __syncthreads();
    mark(9);
    if (tid >= NT - 16) {
        const int ch = (tid >> 2) & 3, h = tid & 3;
        float xc[32];
#pragma unroll
        for (int cc = 0; cc < 32; cc++) xc[cc] = red[(cc * 4 + ch) * 4 + h];
        float t3[4];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) // GGML_F32x8_REDUCE (ggml.c:1354-1371)
            t3[cc] = __fadd_rn(__fadd_rn(__fadd_rn(xc[cc], xc[cc + 16]), __fadd_rn(xc[cc + 8], xc[cc + 24])),
                               __fadd_rn(__fadd_rn(xc[cc + 4], xc[cc + 20]), __fadd_rn(xc[cc + 12], xc[cc + 28])));
        float res = __fadd_rn(__fadd_rn(t3[0], t3[1]), __fadd_rn(t3[2], t3[3]));
#pragma unroll
        for (int jj = 0; jj < 32; jj++)
            if (jj < nleft) res = ps_dot_left(res, lv[jj], lp[jj], jj, nleft); // leftovers, in order (uniform bound)
        if (h < r2) a.att[((int64_t)kvh * r2 + h) * hs + bx * 4 + ch] = res;
    }
    if (dbg) { dbg[10] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

// Batches (prefill chunks, tree verify): one workgroup per (kv head, CI batch columns).  The softmax of a row is computed
// once (not once per channel tile as in the single-token kernel, whose grid has to fill the chip from one column), and a
// V element fetched from L2 serves r2 heads x CI columns.  thread: chain c = tid & 31, channels (tid >> 5) + 32k.
template <int CI>
__global__ __launch_bounds__(PV_NT) void attn_softmax_pv_cols_kernel(psl_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float pl[]; // [CI][r2][n_ctx4] e_j
    const int hs = a.head_size, dim = a.n_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int kvh = blockIdx.x, i0 = blockIdx.y * CI, bs = att_bs(a), pos0 = att_pos0(a);
    const size_t n_kv4 = (size_t)((pos0 + bs + 3) & ~3); // row stride in LDS (attn_softmax_rows)
    __shared__ float redf[R2MAX][PV_NW];
    __shared__ double redd[R2MAX][PV_NW];
    __shared__ float invs[CI][R2MAX];
    int n_kv_c[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ci++) {
        const int i = min(i0 + ci, bs - 1); // (an odd tail repeats the last column; its result is not stored twice)
        n_kv_c[ci] = pos0 + bs; // the mask, not the length, implements causality (executor.cpp:210-224): every row spans n_kv
        attn_softmax_rows(a, i, kvh, r2, pos0, bs, n_kv_c[ci], pl + (size_t)ci * r2 * n_kv4, redf, redd, invs[ci]);
        __syncthreads();
    }
    const int n_kv = pos0 + bs, np = n_kv & ~31;
    const int c = threadIdx.x & 31, dsub = threadIdx.x >> 5;
    float inv[CI][R2MAX];
#pragma unroll
    for (int ci = 0; ci < CI; ci++)
#pragma unroll
        for (int g = 0; g < R2MAX; g++) inv[ci][g] = g < r2 ? invs[ci][g] : 0.f;
    const int dspan = hs / (int)gridDim.z, d_lo = (int)blockIdx.z * dspan; // small grids: the channels are split over blockIdx.z (softmax recomputed)
    for (int d = d_lo + dsub; d < d_lo + dspan; d += PV_NT / 32) {
        const float *vr = a.v_cache + ((int64_t)kvh * hs + d) * a.n_ctx;
        float acc[CI][R2MAX];
#pragma unroll
        for (int ci = 0; ci < CI; ci++)
#pragma unroll
            for (int g = 0; g < R2MAX; g++) acc[ci][g] = 0.f;
        int j = c;
        for (; j + 7 * 32 < np; j += 8 * 32) { // eight V loads in flight per chain
            float vv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) vv[k] = vr[j + 32 * k];
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int ci = 0; ci < CI; ci++)
#pragma unroll
                    for (int g = 0; g < R2MAX; g++)
                        if (g < r2) acc[ci][g] = __fmaf_rn(vv[k], __fmul_rn(pl[(size_t)(ci * r2 + g) * n_kv4 + j + 32 * k], inv[ci][g]), acc[ci][g]);
        }
        for (; j < np; j += 32) {
            const float v = vr[j];
#pragma unroll
            for (int ci = 0; ci < CI; ci++)
#pragma unroll
                for (int g = 0; g < R2MAX; g++)
                    if (g < r2) acc[ci][g] = __fmaf_rn(v, __fmul_rn(pl[(size_t)(ci * r2 + g) * n_kv4 + j], inv[ci][g]), acc[ci][g]);
        }
        const float vleft = (np + c < n_kv) ? vr[np + c] : 0.f; // the n%32 leftovers: one coalesced load, then in order
#pragma unroll
        for (int ci = 0; ci < CI; ci++)
#pragma unroll
            for (int g = 0; g < R2MAX; g++) {
                if (g < r2) {
                    float sres = reduce_f32x8x4(acc[ci][g]);
                    for (int jj = np; jj < n_kv; jj++)
                        sres = ps_dot_left(sres, __shfl(vleft, jj - np, 32), __fmul_rn(pl[(size_t)(ci * r2 + g) * n_kv4 + jj], inv[ci][g]), jj - np, n_kv - np);
                    if (c == 0 && i0 + ci < bs) a.att[(int64_t)(i0 + ci) * dim + ((int64_t)kvh * r2 + g) * hs + d] = sres;
                }
            }
    }
}

// Large batches (prefill chunks): soft-max and V·p as two launches, the second one on the matrix cores.
//  (1) attn_softmax_probs_kernel: grid (n_kv_heads, bs); the r2 score rows of (kv head, column) become probabilities IN PLACE
//      (p_j = e_j * (float)(1/sum), masked positions +0): attn_softmax_rows + ggml_vec_scale_f32.
//  (2) attn_pv_mfma_kernel: out[i][h][d] = ggml_vec_dot_f32(n_kv, V^T[d], p[i][h]).  Chain c of that dot runs over the
//      positions c, c + 32, c + 64, ... in order — a k-ordered fma chain, which is what v_mfma_f32_16x16x4_f32 computes
//      when its accumulator is chained from one instruction to the next: per chain c one accumulator tile of 16 channels
//      x 16 (column, head) pairs, k = four consecutive 32-position blocks per instruction.  After the last block the 32
//      tiles are added in GGML_F32x8_REDUCE's association and the n_kv % 32 leftovers follow one by one (mul, then add).
//      grid (ceil(bs * r2 / 16), n_kv_heads), one wave per 16 head channels.
__global__ __launch_bounds__(PV_NT) void attn_softmax_probs_kernel(psl_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float pl[]; // [r2][n_kv4]
    const int r2 = a.n_heads / a.n_kv_heads, kvh = blockIdx.x, i = blockIdx.y, bs = att_bs(a), pos0 = att_pos0(a);
    const int n_kv = pos0 + bs, n_kv4 = (n_kv + 3) & ~3;
    __shared__ float redf[R2MAX][PV_NW];
    __shared__ double redd[R2MAX][PV_NW];
    __shared__ float invs[R2MAX];
    attn_softmax_rows(a, i, kvh, r2, pos0, bs, n_kv, pl, redf, redd, invs);
    __syncthreads();
    for (int idx = threadIdx.x; idx < r2 * n_kv; idx += PV_NT) {
        const int g = idx / n_kv, j = idx - g * n_kv;
        a.scores[((int64_t)i * a.n_heads + (int64_t)kvh * r2 + g) * a.n_ctx + j] = __fmul_rn(pl[(size_t)g * n_kv4 + j], invs[g]);
    }
}

// (1'), n_kv <= 4096: one WAVE per score row (column i, head h), the row in registers -- no LDS, no barrier.  Lane l holds the
// float4 of elements 4 (64 k + l) ..+3, so a group of 8 of the reference's vector loop (ggml.c:2831-2866) is the lane pair
// (2p, 2p + 1) and its in-group sum tree ((v4 + v0) + (v6 + v2)) + ((v5 + v1) + (v7 + v3)) is one neighbour exchange; the
// n_kv % 8 leftovers take libm's expf on lanes of their own.  grid ceil(bs * n_heads / 4), four rows per workgroup.
constexpr int SMW_MAXQ = 16; // float4 per lane: n_kv <= 64 * 4 * 16
__global__ __launch_bounds__(256) void attn_softmax_probs_wave_kernel(psl_attn_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bs = att_bs(a), pos0 = att_pos0(a), n_kv = pos0 + bs, n8 = n_kv & ~7;
    const int row = (int)blockIdx.x * 4 + wave;
    if (row >= bs * a.n_heads) return;
    const int i = row / a.n_heads;
    float *sr = a.scores + (int64_t)row * a.n_ctx;
    const int nq = (n_kv + 255) >> 8; // float4 trips
    float4 v[SMW_MAXQ];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SMW_MAXQ; k++) {
        if (k < nq) {
            const int j = (k * 64 + lane) * 4;
            const float4 t = j < n_kv ? *(const float4 *)(sr + j) : make_float4(0.f, 0.f, 0.f, 0.f); // (rows are n_ctx long: the float4 stays inside)
            float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int jj = j + r;
                bool ok = false;
                if (jj < n_kv) ok = (jj < pos0) ? (a.kv_vis ? a.kv_vis[jj] != 0 : true) : (a.tree ? a.tree[i * bs + (jj - pos0)] != 0 : (jj - pos0) <= i);
                float x = __fmul_rn(e[r], a.scale);
                x = __fadd_rn(x, ok ? 0.f : -INFINITY);
                e[r] = jj < n_kv ? x : -INFINITY;
                mx = fmaxf(mx, e[r]);
            }
            v[k] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
    mx = wave_max_dpp(mx);
    mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));
    double rs = 0.0;
#pragma unroll
    for (int k = 0; k < SMW_MAXQ; k++) {
        if (k < nq) {
            const int j = (k * 64 + lane) * 4;
            float4 e; // (ps_v_expf_n here measured slower: 12.5 against 10.1 us per launch -- the row's registers, 136, cost a wave per SIMD)
            e.x = ps_v_expf(__fsub_rn(v[k].x, mx)); e.y = ps_v_expf(__fsub_rn(v[k].y, mx));
            e.z = ps_v_expf(__fsub_rn(v[k].z, mx)); e.w = ps_v_expf(__fsub_rn(v[k].w, mx));
            v[k] = e;
            // the neighbour's half of the group of 8 (quad_perm [1, 0, 3, 2])
            const float a0 = __fadd_rn(dpp_f<0xB1>(e.x), e.x), a1 = __fadd_rn(dpp_f<0xB1>(e.y), e.y);
            const float a2 = __fadd_rn(dpp_f<0xB1>(e.z), e.z), a3 = __fadd_rn(dpp_f<0xB1>(e.w), e.w);
            const float gs = __fadd_rn(__fadd_rn(a0, a2), __fadd_rn(a1, a3));
            if (!(lane & 1) && j + 8 <= n8) rs += (double)gs;
        }
    }
    // leftovers n8 .. n_kv - 1: lane t takes element n8 + t
    float et = 0.f;
    const int jt = n8 + lane;
    if (jt < n_kv) {
        const bool ok = (jt < pos0) ? (a.kv_vis ? a.kv_vis[jt] != 0 : true) : (a.tree ? a.tree[i * bs + (jt - pos0)] != 0 : (jt - pos0) <= i);
        float x = __fmul_rn(sr[jt], a.scale);
        x = __fadd_rn(x, ok ? 0.f : -INFINITY);
        et = ps_expf_glibc(__fsub_rn(x, mx));
        rs += (double)et;
    }
    const double tot = wave_sum_d_dpp(rs);
    float inv = (float)(1.0 / tot);
    inv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(inv)));
#pragma unroll
    for (int k = 0; k < SMW_MAXQ; k++) {
        if (k < nq) {
            const int j = (k * 64 + lane) * 4;
            const float4 e = v[k];
            const float4 pq = make_float4(__fmul_rn(e.x, inv), __fmul_rn(e.y, inv), __fmul_rn(e.z, inv), __fmul_rn(e.w, inv));
            if (j + 4 <= n8) *(float4 *)(sr + j) = pq; // (n8 is a multiple of 4: a float4 is wholly inside or wholly leftovers)
        }
    }
    if (jt < n_kv) sr[jt] = __fmul_rn(et, inv);
}

__global__ __launch_bounds__(512, 1) void attn_pv_mfma_kernel(psl_attn_args a) {
    const int hs = a.head_size, dim = a.n_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kvh = blockIdx.y, bs = att_bs(a), n_kv = att_pos0(a) + bs, np = n_kv & ~31, nblk = np >> 5;
    const int rl = lane & 15, m = lane >> 4; // A: row rl (channel), k = m;  B: k = m, column rl
    const int N = bs * r2, n = min((int)blockIdx.x * 16 + rl, N - 1), i = n / r2, g = n - i * r2;
    const int d0 = wave * 16; // (blockDim.x = hs / 16 waves)
    const float *vr = a.v_cache + ((int64_t)kvh * hs + d0 + rl) * a.n_ctx;                     // A: V^T row of channel d0 + rl
    const float *pr = a.scores + ((int64_t)i * a.n_heads + (int64_t)kvh * r2 + g) * a.n_ctx;   // B: probabilities of column n
    ps_f32x4 acc[32];
#pragma unroll
    for (int c = 0; c < 32; c++) acc[c] = ps_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < nblk; b0 += 4) { // four 32-position blocks per instruction: k = m <-> block b0 + m
        const bool in = b0 + m < nblk;
        const float *va = vr + (int64_t)(b0 + m) * 32, *pb = pr + (int64_t)(b0 + m) * 32;
        float av[32], bv[32];
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 t = in ? *(const float4 *)(va + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 w = in ? *(const float4 *)(pb + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
            av[4 * q4] = t.x; av[4 * q4 + 1] = t.y; av[4 * q4 + 2] = t.z; av[4 * q4 + 3] = t.w;
            bv[4 * q4] = w.x; bv[4 * q4 + 1] = w.y; bv[4 * q4 + 2] = w.z; bv[4 * q4 + 3] = w.w;
        }
#pragma unroll
        for (int c = 0; c < 32; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[c], acc[c], 0, 0, 0); // sum = x*y + sum, x = V row
    }
    ps_f32x4 t3[4];
#pragma unroll
    for (int c = 0; c < 4; c++) // GGML_F32x8_REDUCE
        t3[c] = ((acc[c] + acc[c + 16]) + (acc[c + 8] + acc[c + 24])) + ((acc[c + 4] + acc[c + 20]) + (acc[c + 12] + acc[c + 28]));
    ps_f32x4 res = (t3[0] + t3[1]) + (t3[2] + t3[3]);
    // D: column = lane & 15 (the (i, g) pair), rows d0 + 4 * (lane >> 4) + r.  Leftovers: sumf += x[j] * y[j], in order
    const float *vl = a.v_cache + ((int64_t)kvh * hs + d0 + 4 * m) * a.n_ctx;
    for (int jj = np; jj < n_kv; jj++) {
        const float pj = pr[jj];
#pragma unroll
        for (int r = 0; r < 4; r++) res[r] = ps_dot_left(res[r], vl[(int64_t)r * a.n_ctx + jj], pj, jj - np, n_kv - np);
    }
    if ((int)blockIdx.x * 16 + rl < N)
        *(float4 *)(a.att + (int64_t)i * dim + ((int64_t)kvh * r2 + g) * hs + d0 + 4 * m) = make_float4(res[0], res[1], res[2], res[3]);
}

// The same contraction with its operands staged through LDS (head sizes 128 and 64).  The kernel above has every lane
// fetch its own 128-B run of V and of p as eight 16-B loads: a wave instruction touches 64 cache lines for 1 KiB, and the
// CU's address unit, not the matrix core, sets the time (86 us at n_kv = 2048).  Here a wave fetches the 16 x 128 block of
// its V rows with eight row-coalesced instructions (two 512-B rows each), the workgroup fetches the 16 x 128 block of
// probabilities once for all its waves, both are parked in LDS (rows of 128 floats whose 16-byte column index is XOR-ed with row & 7:
// one pass for the operand reads' real lane groups, see attn_scores_mfma_kernel; rows padded to 132 floats took two) one 128-position
// step ahead, and positions past the last full block of 32 are parked as zeros.
constexpr int PVM_RS = 128;
__device__ __forceinline__ size_t pvm_at(const int row, const int seg) { return (size_t)row * PVM_RS + ((seg ^ (row & 7)) << 2); } // float index of 16-byte column seg
template <int NW> // waves per workgroup = head_size / 16
__global__ __launch_bounds__(NW * 64, 1) void attn_pv_mfma_lds_kernel(psl_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float pvs[]; // [2][(NW + 1) * 16][PVM_RS]: the waves' V rows, then the 16 probability rows
    constexpr int hs = NW * 16, NT = NW * 64, ROWS = (NW + 1) * 16, NPS = 512 / NT; // NPS: probability segments per thread
    const int dim = a.n_heads * hs, r2 = a.n_heads / a.n_kv_heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kvh = blockIdx.y, bs = att_bs(a), n_kv = att_pos0(a) + bs, np = n_kv & ~31, nblk = np >> 5, n_it = (nblk + 3) >> 2;
    const int rl = lane & 15, m = lane >> 4;
    const int N = bs * r2, n = min((int)blockIdx.x * 16 + rl, N - 1), i = n / r2, g = n - i * r2;
    const int d0 = wave * 16;
    // staging roles.  V: instruction k covers rows 2k, 2k + 1 of the wave's 16, 32 lanes x 16 B per row
    const int vrow = lane >> 5, seg = lane & 31;
    const float *vg = a.v_cache + ((int64_t)kvh * hs + d0 + vrow) * a.n_ctx; // (the row; the lane's 16-byte segment is added where it is known to exist)
    const float *pg[NPS];
    int prow[NPS];
#pragma unroll
    for (int k = 0; k < NPS; k++) {
        const int sidx = (int)threadIdx.x + k * NT; // segment sidx: row sidx >> 5, 16-B segment sidx & 31
        prow[k] = sidx >> 5;
        const int nn = min((int)blockIdx.x * 16 + prow[k], N - 1), ii = nn / r2, gg = nn - ii * r2;
        pg[k] = a.scores + ((int64_t)ii * a.n_heads + (int64_t)kvh * r2 + gg) * a.n_ctx;
    }
    float4 sv[8], sp[NPS];
    auto fetch = [&](int it) { // (clamped, never branched over: positions past np are zeroed when they are parked.  A segment past np re-reads the ROW's
        // first 16 bytes: with its own column index it left the row -- and, in the last row of a context window of fewer than 128 slots, the allocation)
        const int p0 = it * 128, off = p0 + seg * 4 < np ? p0 + seg * 4 : 0;
#pragma unroll
        for (int k = 0; k < 8; k++) sv[k] = *(const float4 *)(vg + (int64_t)(2 * k) * a.n_ctx + off);
#pragma unroll
        for (int k = 0; k < NPS; k++) {
            const int sg = ((int)threadIdx.x + k * NT) & 31;
            sp[k] = *(const float4 *)(pg[k] + (p0 + sg * 4 < np ? p0 + sg * 4 : 0));
        }
    };
    auto park = [&](int it, float *buf) {
        const int p0 = it * 128;
        const bool vin = p0 + seg * 4 < np;
#pragma unroll
        for (int k = 0; k < 8; k++)
            *(float4 *)(buf + pvm_at(d0 + 2 * k + vrow, seg)) = vin ? sv[k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NPS; k++) {
            const int sg = ((int)threadIdx.x + k * NT) & 31;
            *(float4 *)(buf + pvm_at(NW * 16 + prow[k], sg)) = p0 + sg * 4 < np ? sp[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    ps_f32x4 acc[32];
#pragma unroll
    for (int c = 0; c < 32; c++) acc[c] = ps_f32x4{0.f, 0.f, 0.f, 0.f};
    float *b0p = pvs, *b1p = pvs + (size_t)ROWS * PVM_RS;
    // step `it` computes from buffer it & 1 while the block of step it + 1 (fetched a whole step ago) is parked in the other
    // buffer and the block of step it + 2 is on its way: a fetch has one step of matrix work to hide behind
    if (n_it > 0) { fetch(0); park(0, b0p); fetch(min(1, n_it - 1)); }
    __syncthreads();
    for (int it = 0; it < n_it; it++) {
        float *cur = (it & 1) ? b1p : b0p, *nxt = (it & 1) ? b0p : b1p;
        park(it + 1, nxt); // (past the last step: zeros nobody reads)
        fetch(min(it + 2, n_it - 1));
        float av[32], bv[32];
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 t = *(const float4 *)(cur + pvm_at(d0 + rl, 8 * m + q4)), w = *(const float4 *)(cur + pvm_at(NW * 16 + rl, 8 * m + q4));
            av[4 * q4] = t.x; av[4 * q4 + 1] = t.y; av[4 * q4 + 2] = t.z; av[4 * q4 + 3] = t.w;
            bv[4 * q4] = w.x; bv[4 * q4 + 1] = w.y; bv[4 * q4 + 2] = w.z; bv[4 * q4 + 3] = w.w;
        }
#pragma unroll
        for (int c = 0; c < 32; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[c], acc[c], 0, 0, 0); // sum = x*y + sum, x = V row
        __syncthreads();
    }
    ps_f32x4 t3[4];
#pragma unroll
    for (int c = 0; c < 4; c++) // GGML_F32x8_REDUCE
        t3[c] = ((acc[c] + acc[c + 16]) + (acc[c + 8] + acc[c + 24])) + ((acc[c + 4] + acc[c + 20]) + (acc[c + 12] + acc[c + 28]));
    ps_f32x4 res = (t3[0] + t3[1]) + (t3[2] + t3[3]);
    const float *pr = a.scores + ((int64_t)i * a.n_heads + (int64_t)kvh * r2 + g) * a.n_ctx;
    const float *vl = a.v_cache + ((int64_t)kvh * hs + d0 + 4 * m) * a.n_ctx;
    // leftovers: sumf += x[j] * y[j], in order.  Eight positions' operands are requested before the first one is used (clamped addresses): one at
    // a time every leftover was a memory round trip of its own, up to 31 in a row at the end of every launch
    for (int j0 = np; j0 < n_kv; j0 += 8) {
        float pj[8], vv[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int jj = j0 + k < n_kv ? j0 + k : n_kv - 1;
            pj[k] = pr[jj];
#pragma unroll
            for (int r = 0; r < 4; r++) vv[k][r] = vl[(int64_t)r * a.n_ctx + jj];
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (j0 + k < n_kv) {
#pragma unroll
                for (int r = 0; r < 4; r++) res[r] = ps_dot_left(res[r], vv[k][r], pj[k], j0 + k - np, n_kv - np);
            }
    }
    if ((int)blockIdx.x * 16 + rl < N)
        *(float4 *)(a.att + (int64_t)i * dim + ((int64_t)kvh * r2 + g) * hs + d0 + 4 * m) = make_float4(res[0], res[1], res[2], res[3]);
    // ---- optional: leave the rows quantized for the O projection (saves the quantizer launch between the two).  The workgroup
    // holds 16 (column, head) pairs x hs channels = whole Q8_K tiles of 256 consecutive `att` elements (two heads of one column
    // when hs = 128; psl_attn_pv_quantizes checks the shape): staged through LDS, one wave per tile, ps_quantize_tile as the
    // quantizer kernels run it.
    if (a.qact.qs) {
        float *tl = pvs; // (the last step's barrier is behind every wave)
        *(float4 *)(tl + rl * hs + d0 + 4 * m) = make_float4(res[0], res[1], res[2], res[3]);
        __syncthreads();
        constexpr int TPW = hs * 16 / 256 / NW; // tiles per wave (1 for hs = 128)
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) {
            const int tile = wave * TPW + tt, p0 = tile * (256 / hs); // first pair of the tile
            const int n0 = (int)blockIdx.x * 16 + p0;
            const int nn = min(n0, N - 1), ci = nn / r2, g0 = nn - ci * r2;
            const float4 xv = *(const float4 *)(tl + tile * 256 + lane * 4);
            const float v4[4] = {xv.x, xv.y, xv.z, xv.w};
            const int64_t K = a.qact_K, e = ((int64_t)kvh * r2 + g0) * hs + lane * 4;
            ps_quantize_tile<PS_Q8_K>(v4, n0 < N, e, e / 256, a.qact.qs + (int64_t)ci * K, a.qact.d + (int64_t)ci * (K / 256), a.qact.bs16 + (int64_t)ci * (K / 16),
                                      nullptr, a.qact.qf, ci, K / 256, a.qact.mf);
        }
    }
}

// ---------------------------------------------------------------- fp16-KV decode mode: NOT bit-exact
// ps_hip_model_set_mode bit 3.  The FP32 caches stay the source of truth (prefill, batches, tree verify and every parity
// path read them); K and V are mirrored in fp16, both [n_ctx][kv_dim], and ONLY the single-token attention reads the
// mirrors: half the KV bytes, and — freed from the reference's summation order — a split-KV online soft-max:
//   attn_flash16_kernel   grid (FL_SPLITS, n_kv_heads), 512 threads.  A workgroup takes a contiguous chunk of <= 128
//                         cached positions of one kv head.  Scores: lane = (position, q head of the group), K row as 16 B
//                         loads (the lanes of a position share it), q in registers as fp16 pairs, v_dot2_f32_f16 with fp32
//                         accumulation.  Then per q head one wave: chunk max, p = exp(s - max), chunk sum, and
//                         o[d] = sum_j p_j V[j][d] with lane = channel pair (coalesced V rows).  Writes (o, max, sum).
//                         The workgroup that arrives LAST at its kv head's counter (a relaxed agent-scope fetch_add behind
//                         its drained write-through stores; round 3: the separate combine launch is gone) merges the
//                         FL_SPLITS partials of the group's q heads the usual way.
// Differences from the parity path: fp16 rounding of K, V and q; exp via __expf; fp32 sums in split order.
constexpr int FL_SPLITS = 32, FL_NT = 512;
typedef _Float16 fl_h2 __attribute__((ext_vector_type(2)));
template <int HS> // head_size: 64 or 128
__global__ __launch_bounds__(FL_NT) void attn_flash16_kernel(psl_attn_args a) {
    constexpr int NP = HS / 2;          // fp16 pairs per row
    __shared__ float sc[R2MAX][128];    // scaled, masked scores of the chunk
    const int kvd = a.n_kv_heads * HS, r2 = a.n_heads / a.n_kv_heads;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int n_kv = a.state->pos0 + 1;
    const int chunk = (n_kv + FL_SPLITS - 1) / FL_SPLITS; // <= 128 for n_ctx <= 4096
    const int j0 = split * chunk, j1 = min(j0 + chunk, n_kv), nj = max(j1 - j0, 0);
    // ---- q of the r2 heads as fp16 pairs, through LDS (one conversion per workgroup)
    __shared__ fl_h2 qs[R2MAX][NP];
    for (int t = threadIdx.x; t < r2 * NP; t += FL_NT) {
        const float2 qv = *(const float2 *)(a.q + (int64_t)kvh * r2 * HS + 2 * t);
        fl_h2 h; h.x = (_Float16)qv.x; h.y = (_Float16)qv.y;
        qs[t / NP][t % NP] = h;
    }
    __syncthreads();
    // ---- scores: thread = (position jl, head g)
    {
        const int g = threadIdx.x % r2, jl = threadIdx.x / r2; // r2 in {1, 2, 4, 8}: 512 / r2 >= 64 positions per pass
        fl_h2 qh[NP];
#pragma unroll
        for (int d = 0; d < NP / 4; d++) {
            const uint4 qq = *(const uint4 *)&qs[g][4 * d];
            __builtin_memcpy(&qh[4 * d], &qq.x, 4); __builtin_memcpy(&qh[4 * d + 1], &qq.y, 4); __builtin_memcpy(&qh[4 * d + 2], &qq.z, 4); __builtin_memcpy(&qh[4 * d + 3], &qq.w, 4);
        }
        for (int jb = 0; jb < nj; jb += FL_NT / r2) {
            const int j = j0 + min(jb + jl, nj - 1); // (clamped: every lane loads a valid row, only live ones store)
            const uint4 *kr = (const uint4 *)(a.k16 + (int64_t)j * kvd + kvh * HS);
            uint4 kq[NP / 4];
#pragma unroll
            for (int d = 0; d < NP / 4; d++) kq[d] = kr[d];
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < NP / 4; d++) {
                fl_h2 k0, k1, k2, k3;
                __builtin_memcpy(&k0, &kq[d].x, 4); __builtin_memcpy(&k1, &kq[d].y, 4); __builtin_memcpy(&k2, &kq[d].z, 4); __builtin_memcpy(&k3, &kq[d].w, 4);
                s = __builtin_amdgcn_fdot2(k0, qh[4 * d], s, false);
                s = __builtin_amdgcn_fdot2(k1, qh[4 * d + 1], s, false);
                s = __builtin_amdgcn_fdot2(k2, qh[4 * d + 2], s, false);
                s = __builtin_amdgcn_fdot2(k3, qh[4 * d + 3], s, false);
            }
            if (jb + jl < nj) {
                const bool ok = a.kv_vis ? (j >= n_kv - 1 || a.kv_vis[j] != 0) : true;
                sc[g][jb + jl] = ok ? s * a.scale : -INFINITY;
            }
        }
    }
    __syncthreads();
    // ---- per head: wave g (r2 <= 8 waves)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < r2) {
    const int g = wave, h = kvh * r2 + g;
    float *out = a.part + ((int64_t)h * FL_SPLITS + split) * (HS + 2);
    float s0 = lane < nj ? sc[g][lane] : -INFINITY, s1 = lane + 64 < nj ? sc[g][lane + 64] : -INFINITY;
    const float m = wave_max_dpp(fmaxf(s0, s1));
    const float p0 = (lane < nj && m > -INFINITY) ? __expf(s0 - m) : 0.f, p1 = (lane + 64 < nj && m > -INFINITY) ? __expf(s1 - m) : 0.f;
    float l = p0 + p1;
    l += dpp_f<0xB1>(l); l += dpp_f<0x4E>(l); l += dpp_f<0x141>(l); l += dpp_f<0x140>(l);
    l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), 16)) +
        __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), 48));
    // o: lane owns channels (2 lane, 2 lane + 1) [HS = 128] or lanes < 32 own them [HS = 64]
    float o0 = 0.f, o1 = 0.f;
    const bool own = 2 * lane < HS;
    const _Float16 *vb = a.v16 + (int64_t)j0 * kvd + kvh * HS + 2 * (own ? lane : 0);
    for (int jb = 0; jb < nj; jb += 16) { // sixteen V rows in flight at a time (lanes past the chunk carry p = 0)
        fl_h2 vv[16];
#pragma unroll
        for (int t = 0; t < 16; t++) __builtin_memcpy(&vv[t], vb + (int64_t)min(jb + t, nj - 1) * kvd, 4);
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int jl = jb + t; // uniform
            const float pj = jl < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), jl & 63)) : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), jl & 63));
            o0 = fmaf(pj, (float)vv[t].x, o0);
            o1 = fmaf(pj, (float)vv[t].y, o1);
        }
    }
    // (write-through: the merging workgroup may sit on another XCD, whose L2 never sees this one's plain stores)
    if (own) { coh_store_f(out + 2 * lane, o0); coh_store_f(out + 2 * lane + 1, o1); }
    if (lane == 0) { coh_store_f(out + HS, m); coh_store_f(out + HS + 1, l); }
    }
    // ---- the last workgroup of this kv head to get here merges the partials of its q heads
    __shared__ int is_last;
    __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0): this wave's partials have landed
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.tick + kvh * 64 + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ((t + 1) % FL_SPLITS) == 0;
    }
    __syncthreads();
    if (!is_last) return;
    for (int idx = threadIdx.x; idx < r2 * HS; idx += FL_NT) {
        const int h = kvh * r2 + idx / HS, d = idx % HS;
        const float *pp = a.part + (int64_t)h * FL_SPLITS * (HS + 2);
        float ms[FL_SPLITS], M = -INFINITY;
#pragma unroll
        for (int s = 0; s < FL_SPLITS; s++) { ms[s] = coh_load_f(pp + s * (HS + 2) + HS); M = fmaxf(M, ms[s]); }
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int s = 0; s < FL_SPLITS; s++) {
            const float w = ms[s] > -INFINITY ? __expf(ms[s] - M) : 0.f;
            L = fmaf(w, coh_load_f(pp + s * (HS + 2) + HS + 1), L);
            o = fmaf(w, coh_load_f(pp + s * (HS + 2) + d), o);
        }
        a.att[(int64_t)h * HS + d] = o / L;
    }
}

// ---------------------------------------------------------------- arg-max, first maximum (prob_array.cpp:65-67)
constexpr int AM_PARTS = 64;
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float *src, int64_t n, float *pv, int *pi) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *x = src + (int64_t)blockIdx.y * n;
    const int64_t per = (n + AM_PARTS - 1) / AM_PARTS, lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
        const float v = x[j];
        if (v > best || (v == best && (int)j < idx)) { best = v; idx = (int)j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        pv[blockIdx.y * AM_PARTS + blockIdx.x] = best; pi[blockIdx.y * AM_PARTS + blockIdx.x] = idx;
    }
}
// one wave per row; optionally also performs the greedy-decode bookkeeping (feeds the id back as the next token)
__global__ __launch_bounds__(64) void argmax_final_kernel(const float *pv, const int *pi, int32_t *out, ps_step_state *st,
                                                          int32_t *token, int32_t *ids) {
    const int lane = threadIdx.x;
    float best = pv[blockIdx.x * AM_PARTS + lane]; int idx = pi[blockIdx.x * AM_PARTS + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) {
        if (idx == 0x7fffffff) idx = 0; // a row of NaNs: std::max_element (prob_array.cpp:65-67) returns the first element
        out[blockIdx.x] = idx;
        if (st) { token[0] = idx; ids[st->n_out] = idx; st->n_out += 1; st->pos0 += 1; }
    }
}

} // namespace

void psl_rope_append(hipStream_t st, const psl_attn_args &a, int bs) {
    const int64_t n = (int64_t)bs * (a.n_heads + a.n_kv_heads) * (a.head_size / 2) + (int64_t)bs * a.n_kv_heads * a.head_size;
    int64_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(rope_append_kernel, dim3((unsigned)g), dim3(256), 0, st, a, bs);
}

void psl_attn_scores(hipStream_t st, const psl_attn_args &a, int bs) {
    // workgroups past pos0 + bs exit at once, but a grid sized for n_ctx still costs its dispatch (2048 empty workgroups of the
    // batch kernel: 36 us per call whatever n_kv is) — so eager forwards size the grid with the host's copy of the position
    const int nk = a.n_kv_host > 0 && a.n_kv_host < a.n_ctx ? a.n_kv_host : a.n_ctx;
    if (bs >= 8) { // batches: matrix cores (exact f32 chains), one wave per 16 positions
        dim3 gm((unsigned)((nk + 63) / 64), (unsigned)a.n_kv_heads, (unsigned)SCM_Z);
        if (a.head_size == 128) hipLaunchKernelGGL(attn_scores_mfma_kernel<4>, gm, dim3(256), 0, st, a);
        else if (a.head_size == 64) hipLaunchKernelGGL(attn_scores_mfma_kernel<2>, gm, dim3(256), 0, st, a);
        else if (a.head_size == 32) hipLaunchKernelGGL(attn_scores_mfma_kernel<1>, gm, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn_scores_mfma_kernel<3>, gm, dim3(256), 0, st, a);
        return;
    }
    dim3 g((unsigned)((nk + 31) / 32), (unsigned)a.n_kv_heads, (unsigned)bs);
    if (a.head_size == 128) hipLaunchKernelGGL(attn_scores_kernel<4>, g, dim3(256), 0, st, a);
    else if (a.head_size == 64) hipLaunchKernelGGL(attn_scores_kernel<2>, g, dim3(256), 0, st, a);
    else if (a.head_size == 32) hipLaunchKernelGGL(attn_scores_kernel<1>, g, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_scores_kernel<3>, g, dim3(256), 0, st, a); // head_size 96
}

size_t psl_attn_softmax_pv_lds(const psl_attn_args &a) {
    const int r2 = a.n_heads / a.n_kv_heads;
    return ((size_t)r2 * (((size_t)a.n_ctx + 3) & ~(size_t)3) + 4 * PV_VSTR) * 4;
}
bool psl_attn_pv_quantizes(const psl_attn_args &a, int bs) {
    const int r2 = a.n_heads / a.n_kv_heads;
    return bs >= 8 && a.head_size == 128 && (r2 == 2 || r2 == 4 || r2 == 8) && a.qact.qs != nullptr;
}
void psl_attn_softmax_pv(hipStream_t st, const psl_attn_args &a, int bs) {
    if (bs >= 8 && a.head_size % 16 == 0) { // prefill chunks / wide trees: soft-max in place, then V·p on the matrix cores
        static unsigned long long attrp = 0; // devices that have the attribute
        if (ps_first_on_device(&attrp)) { (void)hipFuncSetAttribute((const void *)attn_softmax_probs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); }
        const int r2p = a.n_heads / a.n_kv_heads;
        const size_t ldsp = (size_t)r2p * (((size_t)a.n_ctx + 3) & ~(size_t)3) * 4;
        if (a.n_ctx <= 64 * 4 * SMW_MAXQ) hipLaunchKernelGGL(attn_softmax_probs_wave_kernel, dim3((unsigned)((bs * a.n_heads + 3) / 4)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn_softmax_probs_kernel, dim3((unsigned)a.n_kv_heads, (unsigned)bs), dim3(PV_NT), ldsp, st, a);
        const dim3 gpv((unsigned)((bs * r2p + 15) / 16), (unsigned)a.n_kv_heads);
        if (a.head_size == 128 || a.head_size == 64) {
            static unsigned long long attrl = 0;
            if (ps_first_on_device(&attrl)) {
                (void)hipFuncSetAttribute((const void *)attn_pv_mfma_lds_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                (void)hipFuncSetAttribute((const void *)attn_pv_mfma_lds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
            }
            if (a.head_size == 128) hipLaunchKernelGGL(attn_pv_mfma_lds_kernel<8>, gpv, dim3(512), 2 * 9 * 16 * PVM_RS * 4, st, a);
            else hipLaunchKernelGGL(attn_pv_mfma_lds_kernel<4>, gpv, dim3(256), 2 * 5 * 16 * PVM_RS * 4, st, a);
            return;
        }
        hipLaunchKernelGGL(attn_pv_mfma_kernel, gpv, dim3((unsigned)(a.head_size / 16 * 64)), 0, st, a);
        return;
    }
    if (bs > 1) { // batches: one workgroup per (kv head, column pair)
        const int r2 = a.n_heads / a.n_kv_heads;
        const size_t row = ((size_t)a.n_ctx + 3) & ~(size_t)3, lds2 = 2 * r2 * row * 4, lds1 = r2 * row * 4;
        static unsigned long long attr2 = 0; // devices that have the attribute
        if (ps_first_on_device(&attr2)) {
            (void)hipFuncSetAttribute((const void *)attn_softmax_pv_cols_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
            (void)hipFuncSetAttribute((const void *)attn_softmax_pv_cols_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        }
        // few (kv head, column pair) workgroups (tree verify, prefill tails): split the head channels over blockIdx.z
        const bool two = lds2 <= 150 * 1024;
        const int wgs = a.n_kv_heads * (two ? (bs + 1) / 2 : bs);
        unsigned nz = 1;
        while (nz * 32 < (unsigned)a.head_size && wgs * (int)nz * 2 <= 256 && a.head_size % (nz * 2 * 32) == 0) nz *= 2;
        if (two) hipLaunchKernelGGL(attn_softmax_pv_cols_kernel<2>, dim3((unsigned)a.n_kv_heads, (unsigned)((bs + 1) / 2), nz), dim3(PV_NT), lds2, st, a);
        else hipLaunchKernelGGL(attn_softmax_pv_cols_kernel<1>, dim3((unsigned)a.n_kv_heads, (unsigned)bs, nz), dim3(PV_NT), lds1, st, a);
        return;
    }
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) { (void)hipFuncSetAttribute((const void *)attn_softmax_pv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); }
    dim3 g((unsigned)(a.head_size / 4), (unsigned)a.n_kv_heads, (unsigned)bs);
    hipLaunchKernelGGL(attn_softmax_pv_kernel, g, dim3(PV_NT), psl_attn_softmax_pv_lds(a), st, a);
}

bool psl_attn_decode_f16(hipStream_t st, const psl_attn_args &a) {
    if (!a.k16 || !a.v16 || !a.part || !a.tick || a.tree || a.n_ctx > 128 * FL_SPLITS) return false;
    const int r2 = a.n_heads / a.n_kv_heads;
    if ((r2 != 1 && r2 != 2 && r2 != 4 && r2 != 8) || (a.head_size != 128 && a.head_size != 64)) return false;
    const dim3 g(FL_SPLITS, (unsigned)a.n_kv_heads);
    if (a.head_size == 128) hipLaunchKernelGGL(attn_flash16_kernel<128>, g, dim3(FL_NT), 0, st, a);
    else hipLaunchKernelGGL(attn_flash16_kernel<64>, g, dim3(FL_NT), 0, st, a);
    return true;
}

// single token, one launch (attn_decode2_kernel); false: not covered, the caller launches scores + softmax / V.p
size_t psl_attn_decode2_xchg_bytes(int n_kv_heads, int n_ctx) { return (size_t)n_kv_heads * 4 * (((size_t)n_ctx + 31) & ~(size_t)31) * 4; }
bool psl_attn_decode2(hipStream_t st, int n_cu, const psl_attn_args &a) {
    const int r2 = a.n_heads / a.n_kv_heads, gx = a.head_size / 4;
    if (!a.xchg || !a.tick || !a.sync || a.tree || r2 > 4 || a.n_kv_heads > 32 || a.n_ctx > D2_MAXCTX || (a.head_size != 128 && a.head_size != 64)) return false;
    if (gx * a.n_kv_heads > n_cu) return false; // every workgroup resident (one per CU at n_ctx = 4096)
    const int RS = ((a.n_ctx + 127) & ~127) + 36;
    const size_t lds = ((size_t)8 * RS + 512 + 128 + 16 + 32 + 4 * a.head_size + (16 * 32 / (a.head_size / 32)) * 4 + 32 + 64) * 4;
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) {
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<4, 1024, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<4, 1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<2, 1024, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<2, 1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
    const dim3 g((unsigned)(gx * a.n_kv_heads));
    // 512-thread workgroups (template parameter NT): a kernel boundary behind them costs ~1.5 us less than behind 1024-thread ones.  Round 3 measured them
    // slower in total (14.05 vs 13.66 us: twice the soft-max numerators per lane, issue-bound); with ps_v_expf_n they win: 8B decode 534.9 -> 542.8 tok/s
    // (profiles/r04_gemv_variants.txt).  PS_ATTN_NT=1024 brings the round-3 geometry back for an A/B run.
    static const int nt_env = getenv("PS_ATTN_NT") ? atoi(getenv("PS_ATTN_NT")) : 512;
    static unsigned long long attr5 = 0;
    if (ps_first_on_device(&attr5)) {
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<4, 512, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<4, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<2, 512, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void *)attn_decode2_kernel<2, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
    if (nt_env == 512) {
        if (a.head_size == 128) { if (a.kv_stream) hipLaunchKernelGGL((attn_decode2_kernel<4, 512, 1>), g, dim3(512), lds, st, a); else hipLaunchKernelGGL((attn_decode2_kernel<4, 512, 0>), g, dim3(512), lds, st, a); }
        else { if (a.kv_stream) hipLaunchKernelGGL((attn_decode2_kernel<2, 512, 1>), g, dim3(512), lds, st, a); else hipLaunchKernelGGL((attn_decode2_kernel<2, 512, 0>), g, dim3(512), lds, st, a); }
        return true;
    }
    if (a.head_size == 128) { if (a.kv_stream) hipLaunchKernelGGL((attn_decode2_kernel<4, 1024, 1>), g, dim3(1024), lds, st, a); else hipLaunchKernelGGL((attn_decode2_kernel<4, 1024, 0>), g, dim3(1024), lds, st, a); }
    else { if (a.kv_stream) hipLaunchKernelGGL((attn_decode2_kernel<2, 1024, 1>), g, dim3(1024), lds, st, a); else hipLaunchKernelGGL((attn_decode2_kernel<2, 1024, 0>), g, dim3(1024), lds, st, a); }
    return true;
}

void psl_argmax2(hipStream_t st, const float *src, int64_t n, int64_t rows, int32_t *out, float *part_v, int *part_i,
                 ps_step_state *state, int32_t *token, int32_t *ids) {
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(AM_PARTS, (unsigned)rows), dim3(256), 0, st, src, n, part_v, part_i);
    hipLaunchKernelGGL(argmax_final_kernel, dim3((unsigned)rows), dim3(64), 0, st, part_v, part_i, out, state, token, ids);
}
