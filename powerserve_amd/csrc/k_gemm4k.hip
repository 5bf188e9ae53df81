// Q4_K batched mat-mul (prefill chunks, tree verify) on v_mfma_i32_16x16x32_i8 -- bit-exact with ggml_vec_dot_q4_K_q8_K.
//
// The AVX2 kernel (libs/ggml/src/ggml-quants.c:7809-7873) keeps, per super-block of 256 weights, ONE int32 per accumulator
// lane u:  sumi[u] = sum over the 8 sub-blocks g of  scale[g] * dot4(q[g][4u..4u+3], y[g][4u..4u+3]),  accumulated in
// int32 before the single  acc[u] = fma(d * y.d, (float)sumi[u], acc[u]).  Integers associate, and the 32 elements lane u
// owns in a super-block are exactly one K = 32 contraction: for a tile of 16 weight rows x 16 activation columns and one u,
//     D[row][col] = sum_k A[row][k] * B[k][col],   k = (g, e),  A = q4 * factor,  B = y
// is one v_mfma_i32_16x16x32_i8.  The 6-bit scale cannot ride in an int8 operand whole (15 * 63 > 127), so it is split,
// scale = 8 * hi + lo with 3-bit halves (q * 7 <= 105):  sumi = 8 * (A_hi . B) + (A_lo . B)  -- two MFMAs, the first result
// shifted into the C input of the second.  The mins term (ggml-quants.c:7831-7834) per accumulator lane v is
//     prod[v] = mins[2v] * q8sum[2v] + mins[2v+1] * q8sum[2v+1]
// -- a K = 4 contraction over the four 16-sums of the two sub-blocks, every factor an integer fp16 holds exactly (mins <= 63,
// |16-sum| <= 2032), every partial sum below 2^24: v_mfma_f32_16x16x16_f16 returns (float)prod itself.  Everything after
// sumi / prod is the reference's fp32 arithmetic per (row, column, lane): the eight acc chains, the four acc_m chains,
// hsum_float_8.
//
// Operand layout (tools/micro/mfma32probe.hip): lane l supplies A[i = l % 16][k = 8 * (l / 16) ..+7] and
// B[k = 8 * (l / 16) ..+7][j = l % 16]; result register r of lane l is D[i = 4 * (l / 16) + r][j = l % 16].  With
// k = 8 * kb + 4 * half + e  <->  sub-block g = 2 * kb + half, element 4u + e:
//   * A: the lane's 8 bytes come from ONE dword of the lane-major weight layout (ps_internal.h): byte 32 kb + 4u + e of
//     the super-block holds element e of sub-block 2 kb in its low nibble and of sub-block 2 kb + 1 in its high nibble;
//   * B: the quantizer writes a second, fragment-major copy of the Q8_K quants (ps_act::qf): per (16 columns,
//     super-block) 4 KiB laid out [u / 2][lane][u % 2][half][4 B], so that a wave's B operands for two values of u are
//     one fully coalesced 1 KiB load; the column metadata rides tile-major next to it (ps_act::mf).
//
// A workgroup is twelve waves on 16 weight rows x 128 columns.  Waves 0-7 COMPUTE: one 16 x 16 tile each (48 fp32 chains
// per lane), walking K.  Waves 8-11 PRODUCE: everything the eight computing waves would otherwise each derive from the
// same 16 rows -- the nibbles times the split scales as ready MFMA A operands, the mins as fp16 A operands, d and dmin as
// fp32 -- is made ONCE per super-block by the producers (four rows each) and parked in LDS, one step ahead of the
// consumers, one barrier per step.  The producers own their memory pipeline: a register ring of G4K_RING super-blocks of
// their rows is in flight from HBM, so nobody waits on a weight load (what-if, round 2: the per-wave version spent a
// third of its vector instructions on these shared derivations, tools/g4k_exp.py).
// EPI 1 (SiLU(gate) * up) runs the gate tile's K loop, keeps its four results, then the up tile's.
#include "ps_gemv_dev.h"

namespace {

typedef int g4k_i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 g4k_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 g4k_h4 __attribute__((ext_vector_type(4)));
typedef float g4k_f4 __attribute__((ext_vector_type(4)));

struct G4KMat {
    const uint8_t *qs, *aux;
    float *out;
    const float *bias;
    int64_t N, ldo;
    int n_tiles; // N / 16
};
struct G4KParams {
    G4KMat w[3];
    int n_w, nsb, bs, n_tasks;
    const float *residual;
    const int8_t *qf;   // fragment-major quants
    const uint8_t *mf;  // tile-major column metadata (ps_act::mf)
    unsigned long long *dbg; // timeline slots (ps_hip_debug_timeline keys 48..50, 52), or null
};

__device__ __forceinline__ long g4k_pack(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | lo); }

// the fp32 chains of one 16 x 16 tile: rows 4 kb + r of the tile, this lane's column
struct G4KAcc {
    float acc[4][8], accm[4][4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int u = 0; u < 8; u++) acc[r][u] = 0.f;
#pragma unroll
            for (int v = 0; v < 4; v++) accm[r][v] = 0.f;
        }
    }
    // hsum_float_8 (ggml-quants.c:62-68) and the acc_m reduction, as row_reduce<PS_Q4_K> does with lane shifts
    __device__ __forceinline__ void reduce(float (&y)[4]) const {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float s[4];
#pragma unroll
            for (int k = 0; k < 4; k++) s[k] = __fadd_rn(acc[r][k], acc[r][k + 4]);
            const float res = __fadd_rn(__fadd_rn(s[0], s[2]), __fadd_rn(s[1], s[3]));
            const float mm  = __fadd_rn(__fadd_rn(accm[r][0], accm[r][2]), __fadd_rn(accm[r][1], accm[r][3]));
            y[r] = __fadd_rn(res, mm);
        }
    }
};

// One LDS stage = one super-block of the workgroup's 16 rows, as the consumers want it:
//   [row][u][kb] 16 B = (A_hi, A_lo) of lane (row, kb) for accumulator lane u; rows padded to 528 B (b128 reads of 16 rows
//                       at one kb hit 16 distinct 16-B bank groups)
//   [row] 32 B   = the mins as the four fp16 A operands (m[2v], m[2v], m[2v+1], m[2v+1]), v = 0..3
//   [row] 8 B    = (d, dmin) as fp32
constexpr int G4K_RS = 528, G4K_MINS = 16 * G4K_RS, G4K_DD = G4K_MINS + 16 * 32, G4K_STAGE = G4K_DD + 16 * 8;
constexpr int G4K_NC = 8, G4K_NP = 4, G4K_RING = 4; // computing waves, producing waves, super-blocks in flight per producer

// ---- producers.  Wave h (0..3) owns rows 4h .. 4h+3 of the tile; lane = (row 4h + l / 16, u = (l / 2) % 8, kb pair l % 2):
// its two weight dwords [row][u][kb = 2p, 2p + 1] are one 8-B load (128 lanes-worth: the 512 contiguous bytes of 4 rows).
// stages 0 .. nsb-1: tile of (qs0, aux0); stages nsb .. n_stages-1 (EPI 1): the same tile of (qs1, aux1)
__device__ __forceinline__ void g4k_produce(const uint2 q, const uint4 h, char *st, const int row, const int u, const int p, const int lane) {
    // the four sub-block scales 4p .. 4p+3 (get_scale_min_k4: 0..3 sit in the low 6 bits of scale bytes 0..3, 4..7 are
    // spread over bytes 8..11 and the top bits of bytes 0..3), split into 3-bit halves, each replicated to a 16-bit pair
    const uint32_t scb = p ? ((h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4)) : (h.y & 0x3f3f3f3fu);
    const uint32_t hi4 = (scb >> 3) & 0x07070707u, lo4 = scb & 0x07070707u;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pkmul = [](uint32_t a, uint32_t f) { u16x2 va, vf; __builtin_memcpy(&va, &a, 4); __builtin_memcpy(&vf, &f, 4); va = va * vf; uint32_t o; __builtin_memcpy(&o, &va, 4); return o; };
#pragma unroll
    for (int e = 0; e < 2; e++) { // kb = 2p + e: sub-blocks 2 kb (low nibbles), 2 kb + 1 (high nibbles) = bytes 2e, 2e + 1 of scb
        const uint32_t w = e ? q.y : q.x;
        const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;
        const uint32_t s0 = 0x0c000c00u | (uint32_t)(2 * e) * 0x00010001u, s1 = 0x0c000c00u | (uint32_t)(2 * e + 1) * 0x00010001u;
        const uint32_t f0h = __builtin_amdgcn_perm(0u, hi4, s0), f1h = __builtin_amdgcn_perm(0u, hi4, s1);
        const uint32_t f0l = __builtin_amdgcn_perm(0u, lo4, s0), f1l = __builtin_amdgcn_perm(0u, lo4, s1);
        *(uint4 *)(st + row * G4K_RS + u * 64 + (2 * p + e) * 16) = make_uint4(pkmul(lo, f0h), pkmul(hi, f1h), pkmul(lo, f0l), pkmul(hi, f1l));
    }
    if ((lane & 15) == 0) { // once per row: mins and (d, dmin)
        const uint32_t mn03 = h.z & 0x3f3f3f3fu;
        const uint32_t mn47 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
        uint32_t o[8];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const uint32_t mp = (v < 2) ? mn03 : mn47;
            const int e = (2 * v) & 3;
            // bytes (m, 0x64, m, 0x64) = the fp16 pair (1024 + m, 1024 + m); minus 1024 is exact
            const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, mp, 0x04000400u | (uint32_t)(e * 0x00010001u));
            const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, mp, 0x04000400u | (uint32_t)((e + 1) * 0x00010001u));
            g4k_h2 h0, h1;
            __builtin_memcpy(&h0, &p0, 4); __builtin_memcpy(&h1, &p1, 4);
            const g4k_h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
            h0 = h0 - k1024; h1 = h1 - k1024;
            __builtin_memcpy(&o[2 * v], &h0, 4); __builtin_memcpy(&o[2 * v + 1], &h1, 4);
        }
        *(uint4 *)(st + G4K_MINS + row * 32) = make_uint4(o[0], o[1], o[2], o[3]);
        *(uint4 *)(st + G4K_MINS + row * 32 + 16) = make_uint4(o[4], o[5], o[6], o[7]);
        *(float2 *)(st + G4K_DD + row * 8) = make_float2(ps_h2f((uint16_t)(h.x & 0xffff)), ps_h2f((uint16_t)(h.x >> 16)));
    }
}

__device__ __forceinline__ void g4k_producer_wave(const uint8_t *qs0, const uint8_t *aux0, const uint8_t *qs1, const uint8_t *aux1, const int tile,
                                                  const int nsb, const int n_stages, char *lds, const int hw, unsigned long long *dbg) {
    const int lane = threadIdx.x & 63;
    int dbg_n = 1;
    auto mark = [&](int g) { if (dbg && (g < 8 || (g & 7) == 7) && dbg_n < 29) dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); };
    const int row = 4 * hw + (lane >> 4), u = (lane >> 1) & 7, p = lane & 1, unit = row >> 3, r8 = row & 7;
    const size_t qo = ((size_t)(2 * tile + unit) * nsb << 10) + (size_t)(r8 * 32 + u * 4 + 2 * p) * 4;
    const size_t ho = (size_t)(2 * tile + unit) * nsb * 128 + (size_t)r8 * 16;
    auto ldq = [&](int g) -> uint2 { // (stages past the end are clamped, never branched over)
        g = g < n_stages ? g : n_stages - 1;
        const uint8_t *b = g < nsb ? qs0 : qs1;
        return *(const uint2 *)(b + qo + ((size_t)(g < nsb ? g : g - nsb) << 10));
    };
    auto ldh = [&](int g) -> uint4 {
        g = g < n_stages ? g : n_stages - 1;
        const uint8_t *b = g < nsb ? aux0 : aux1;
        return *(const uint4 *)(b + ho + (size_t)(g < nsb ? g : g - nsb) * 128);
    };
    uint2 rq[G4K_RING];
    uint4 rh[G4K_RING];
#pragma unroll
    for (int k = 0; k < G4K_RING; k++) { rq[k] = ldq(k); rh[k] = ldh(k); }
    for (int g0 = 0; g0 < n_stages; g0 += G4K_RING) { // (n_stages % G4K_RING == 0: psk_gemm4k checks)
#pragma unroll
        for (int k = 0; k < G4K_RING; k++) {
            const uint2 q = rq[k];
            const uint4 h = rh[k];
            rq[k] = ldq(g0 + k + G4K_RING); rh[k] = ldh(g0 + k + G4K_RING);
            g4k_produce(q, h, lds + ((g0 + k) & 3) * G4K_STAGE, row, u, p, lane);
            if (k & 1) __syncthreads(); // one barrier per PAIR of stages: stages g0 + k - 1, g0 + k are parked
            mark(g0 + k);
        }
    }
}

// ---- consumers: one super-block of one tile from the parked operands
__device__ __forceinline__ void g4k_superblock(G4KAcc &T, const char *st, const char *zero, const ps_u32x4 (&bq)[4], const float yd,
                                               const ps_u32x4 b16a, const ps_u32x4 b16b, const int m, const int kb) {
    const g4k_f4 dda = *(const g4k_f4 *)(st + G4K_DD + kb * 32), ddb = *(const g4k_f4 *)(st + G4K_DD + kb * 32 + 16); // (d, dmin) of rows 4 kb + r
    const float dr[4] = {__fmul_rn(yd, dda[0]), __fmul_rn(yd, dda[2]), __fmul_rn(yd, ddb[0]), __fmul_rn(yd, ddb[2])};
    const float dmin[4] = {__fmul_rn(-yd, dda[1]), __fmul_rn(-yd, dda[3]), __fmul_rn(-yd, ddb[1]), __fmul_rn(-yd, ddb[3])};
    const char *ap = st + m * G4K_RS + kb * 16;
    // the MFMAs in rounds of four independent ones (the second round of a group takes the first round's results, shifted,
    // as its C input): the matrix core's latency is covered by the other three, not by wait states
#pragma unroll
    for (int uh = 0; uh < 8; uh += 4) {
        g4k_i32x4 cc[4];
        long al[4], bb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int u = uh + k;
            const uint4 ao = *(const uint4 *)(ap + u * 64);
            al[k] = g4k_pack(ao.z, ao.w);
            const uint32_t b0 = (u & 1) ? bq[u >> 1].z : bq[u >> 1].x, b1 = (u & 1) ? bq[u >> 1].w : bq[u >> 1].y;
            bb[k] = g4k_pack(b0, b1);
            const g4k_i32x4 z = {0, 0, 0, 0};
            cc[k] = __builtin_amdgcn_mfma_i32_16x16x32_i8(g4k_pack(ao.x, ao.y), bb[k], z, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) cc[k] = __builtin_amdgcn_mfma_i32_16x16x32_i8(al[k], bb[k], cc[k] << 3, 0, 0, 0); // sumi[u] of rows 4 kb + r, this lane's column
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int r = 0; r < 4; r++) T.acc[r][uh + k] = __fmaf_rn(dr[r], (float)cc[k][r], T.acc[r][uh + k]);
        }
    }
    // acc_m: the lanes of k-group 0 supply the row's mins operands, the others zeros; B = the column's fp16 16-sums
    const char *mp = kb == 0 ? st + G4K_MINS + m * 32 : zero;
    const uint4 ma = *(const uint4 *)mp, mb = *(const uint4 *)(mp + 16);
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const uint32_t ax = v == 0 ? ma.x : v == 1 ? ma.z : v == 2 ? mb.x : mb.z, ay = v == 0 ? ma.y : v == 1 ? ma.w : v == 2 ? mb.y : mb.w;
        const uint32_t bx = v == 0 ? b16a.x : v == 1 ? b16a.z : v == 2 ? b16b.x : b16b.z;
        const uint32_t by = v == 0 ? b16a.y : v == 1 ? b16a.w : v == 2 ? b16b.y : b16b.w;
        g4k_h2 a0, a1, g0, g1;
        __builtin_memcpy(&a0, &ax, 4); __builtin_memcpy(&a1, &ay, 4); __builtin_memcpy(&g0, &bx, 4); __builtin_memcpy(&g1, &by, 4);
        const g4k_h4 am = {a0[0], a0[1], a1[0], a1[1]}, bm = {g0[0], g0[1], g1[0], g1[1]};
        const g4k_f4 zf = {0.f, 0.f, 0.f, 0.f};
        const g4k_f4 pr = __builtin_amdgcn_mfma_f32_16x16x16f16(am, bm, zf, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) T.accm[r][v] = __fmaf_rn(dmin[r], pr[r], T.accm[r][v]);
    }
}

// one tile of the consumers' walk: `s_first` = index of its first step in the workgroup's stage sequence (LDS stage = step % 2)
__device__ __forceinline__ void g4k_tile(const int nsb, const int8_t *qf_ct, const uint8_t *mf_ct, const int mc, const char *lds, const char *zero,
                                         const int s_first, float (&y)[4], unsigned long long *dbg, int &dbg_n) {
    const int lane = threadIdx.x & 63, m = lane & 15, kb = lane >> 4;
    auto mark = [&](int g) { if (dbg && (g < 8 || (g & 7) == 7) && dbg_n < 29) dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); };
    G4KAcc T;
    T.clear();
    ps_u32x4 bn[4]; // the B fragments of the next super-block (L2: a few hundred cycles -- one step ahead is enough)
#pragma unroll
    for (int up = 0; up < 4; up++) bn[up] = *(const ps_u32x4 *)(qf_ct + up * 1024 + lane * 16);
    for (int sb = 0; sb < nsb; sb++) {
        const ps_u32x4 bq[4] = {bn[0], bn[1], bn[2], bn[3]};
        const uint8_t *mfs = mf_ct + (size_t)sb * 576; // the (column tile, super-block) block: d[16], then the fp16 16-sums [16][16]
        const float yd = *(const float *)(mfs + mc * 4);
        const ps_u32x4 b16a = *(const ps_u32x4 *)(mfs + 64 + mc * 32), b16b = *(const ps_u32x4 *)(mfs + 64 + mc * 32 + 16);
        {
            const int nb = sb + 1 == nsb ? 0 : sb + 1; // (the following tile of an EPI 1 pair meets the same columns from super-block 0)
#pragma unroll
            for (int up = 0; up < 4; up++) bn[up] = *(const ps_u32x4 *)(qf_ct + ((size_t)nb << 12) + up * 1024 + lane * 16);
        }
        if (!(sb & 1)) __syncthreads(); // (s_first is even) the producers have parked this step and the next
        mark(s_first + sb);
        g4k_superblock(T, lds + ((s_first + sb) & 3) * G4K_STAGE, zero, bq, yd, b16a, b16b, m, kb);
    }
    T.reduce(y);
}

// twelve waves: the eight column tiles of ONE row task (128 columns per workgroup) + the four producers
template <int EPI>
__global__ __launch_bounds__((G4K_NC + G4K_NP) * 64) void gemm4k_kernel(const G4KParams p) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    // wave -> (row task, column tile)
    const int ct = (int)blockIdx.y * 8 + (wave & 7);
    const int task = (int)blockIdx.x; // (grid.x = tasks exactly; every wave stays for the barriers)
    __shared__ __attribute__((aligned(16))) char lds[4 * G4K_STAGE + 32];
    if (threadIdx.x < 8) ((uint32_t *)(lds + 4 * G4K_STAGE))[threadIdx.x] = 0u; // the zero operands (visible after barrier #0)
    int wi = 0, tile = task;
    if (EPI != 1) {
        if (p.n_w > 1 && tile >= p.w[0].n_tiles) { tile -= p.w[0].n_tiles; wi = 1; }
        if (p.n_w > 2 && wi == 1 && tile >= p.w[1].n_tiles) { tile -= p.w[1].n_tiles; wi = 2; }
    }
    const G4KMat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
    // timeline: consumer wave 0 -> words 0..31, producer wave 8 -> 32..63 of the workgroup's slot ([0]/[31] entry / exit clock, [29]/[30] 100 MHz)
    unsigned long long *const dbg = (p.dbg && blockIdx.y == 0 && blockIdx.x < 1024 && lane == 0 && (wave == 0 || wave == G4K_NC)) ? p.dbg + ((size_t)blockIdx.x * 2 + (wave == G4K_NC)) * 32 : nullptr;
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }
    if (wave >= G4K_NC) {
        g4k_producer_wave(W.qs, W.aux, EPI == 1 ? p.w[1].qs : W.qs, EPI == 1 ? p.w[1].aux : W.aux, tile, p.nsb, EPI == 1 ? 2 * p.nsb : p.nsb, lds, wave - G4K_NC, dbg);
        if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
        return;
    }
    const int col = ct * 16 + m, colc = col < p.bs ? col : p.bs - 1; // (the last tile may be ragged: clamp the column metadata)
    const int ctc = ct * 16 < p.bs ? ct : (p.bs - 1) / 16; // (a wave past the batch walks the last tile's columns and stores nothing)
    const int8_t *qf_ct = p.qf + ((size_t)ctc * p.nsb << 12);
    const uint8_t *mf_ct = p.mf + (size_t)ctc * p.nsb * 576;
    const int mc = colc & 15;
    const char *zero = lds + 4 * G4K_STAGE;
    float y[4];
    int dbg_n = 1;
    g4k_tile(p.nsb, qf_ct, mf_ct, mc, lds, zero, 0, y, dbg, dbg_n);
    if (EPI == 1) {
        float yu[4];
        g4k_tile(p.nsb, qf_ct, mf_ct, mc, lds, zero, p.nsb, yu, dbg, dbg_n);
#pragma unroll
        for (int r = 0; r < 4; r++) y[r] = ps_silu_mul(y[r], yu[r]);
    }
    if (col < p.bs) {
        const int64_t row0 = (int64_t)tile * 16 + kb * 4;
        float *o = W.out + (int64_t)col * W.ldo + row0;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = y[r];
            if (EPI != 1) {
                if (W.bias) v[r] = __fadd_rn(v[r], W.bias[row0 + r]);
                if (p.residual && wi == 0) v[r] = __fadd_rn(p.residual[(int64_t)col * W.ldo + row0 + r], v[r]);
            }
        }
        *(float4 *)o = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

} // namespace

// Q4_K batched mat-mul from fragment-major Q8_K activations (act.qf).  -1: not covered (the caller takes gemm8m).
int psk_gemm4k(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K, int64_t bs) {
    static const bool off = getenv("PS_NO_GEMM4K") != nullptr; // (A/B switch for measurements)
    if (off || a.pro != 0 || a.rope || a.n_w < 1 || !act.qf || K % 256) return -1;
    G4KParams p{};
    int tiles_total = 0;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != PS_Q4_K || a.w[i]->K != K || a.w[i]->N % 16 || a.ldo[i] % 4) return -1;
        p.w[i] = G4KMat{a.w[i]->qs, a.w[i]->aux, a.out[i], a.bias[i], a.w[i]->N, a.ldo[i], (int)(a.w[i]->N / 16)};
        tiles_total += p.w[i].n_tiles;
    }
    const int epi = a.silu_pair ? 1 : 0;
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N || a.ldo[0] != a.ldo[1])) return -1;
    p.n_w = a.n_w; p.nsb = (int)(K / 256); p.bs = (int)bs;
    p.n_tasks = epi == 1 ? p.w[0].n_tiles : tiles_total;
    p.residual = a.residual; p.qf = act.qf; p.mf = act.mf;
    p.dbg = psk_gemv_dbg_buf(12 + epi, epi ? 0 : (a.n_w == 3 ? 0 : (K <= 8192 ? 1 : 2))); // keys 48 QKV, 49 O, 50 down, 52 gate/up
    const int n_ct = (int)((bs + 15) / 16);
    // Below eight column tiles a workgroup's waves would not share their weight rows any more, and the kernels that spread
    // a row group's integer work over producer waves (gemm8) are ahead there: tree forward of the 8B shape, ms by width,
    // this kernel / gemm8: 2: 12.5 / 4.9, 12: 13.7 / 6.4, 32: 14.4 / 9.2, 64: 18.3 / 13.3, 128: 17.0 / 22.9
    // (profiles/r02_tree_forward_latency_8b.json).
    if (n_ct < 8 || p.nsb % 4) return -1;
    (void)n_cu;
    const dim3 grid((unsigned)p.n_tasks, (unsigned)((n_ct + 7) / 8));
    static_assert(G4K_RING == 4, "nsb % 4 == 0 is what the producers' ring is unrolled for");
    if (epi == 1) hipLaunchKernelGGL((gemm4k_kernel<1>), grid, dim3((G4K_NC + G4K_NP) * 64), 0, st, p);
    else hipLaunchKernelGGL((gemm4k_kernel<0>), grid, dim3((G4K_NC + G4K_NP) * 64), 0, st, p);
    return 0;
}
