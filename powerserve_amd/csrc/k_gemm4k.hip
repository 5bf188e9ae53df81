// Q4_K batched mat-mul (prefill chunks, tree verify) on v_mfma_i32_16x16x32_i8 -- bit-exact with ggml_vec_dot_q4_K_q8_K.
//
// The AVX2 kernel (libs/ggml/src/ggml-quants.c:7809-7873) keeps, per super-block of 256 weights, ONE int32 per accumulator
// lane u:  sumi[u] = sum over the 8 sub-blocks g of  scale[g] * dot4(q[g][4u..4u+3], y[g][4u..4u+3]),  accumulated in
// int32 before the single  acc[u] = fma(d * y.d, (float)sumi[u], acc[u]).  Integers associate, and the 32 elements lane u
// owns in a super-block are exactly one K = 32 contraction: for a tile of 16 weight rows x 16 activation columns and one u,
//     D[row][col] = sum_k A[row][k] * B[k][col],   k = (g, e),  A = q4 * factor,  B = y
// is one v_mfma_i32_16x16x32_i8.  The 6-bit scale cannot ride in an int8 operand whole (15 * 63 > 127), so it is split,
// scale = 8 * hi + lo with 3-bit halves (q * 7 <= 105):  sumi = 8 * (A_hi . B) + (A_lo . B)  -- two MFMAs, the first result
// shifted into the C input of the second.  Everything after sumi is the reference's fp32 arithmetic, per (row, column, u):
// the eight acc chains, the four acc_m chains of the mins (ggml-quants.c:7831-7834), hsum_float_8.
//
// Operand layout (tools/micro/mfma32probe.hip): lane l supplies A[i = l % 16][k = 8 * (l / 16) ..+7] and
// B[k = 8 * (l / 16) ..+7][j = l % 16]; result register r of lane l is D[i = 4 * (l / 16) + r][j = l % 16].  With
// k = 8 * kb + 4 * half + e  <->  sub-block g = 2 * kb + half, element 4u + e:
//   * A: the lane's 8 bytes come from ONE dword of the lane-major weight layout (ps_internal.h): byte 32 kb + 4u + e of
//     the super-block holds element e of sub-block 2 kb in its low nibble and of sub-block 2 kb + 1 in its high nibble;
//   * B: the quantizer writes a second, fragment-major copy of the Q8_K quants (ps_act::qf): per (16 columns,
//     super-block) 4 KiB laid out [u / 2][lane][u % 2][half][4 B], so that a wave's B operands for two values of u are
//     one fully coalesced 1 KiB load.
// A workgroup is eight waves; a wave owns ONE 16 x 16 tile at a time (48 fp32 chains per lane) and walks K; the waves of a
// workgroup take the eight column tiles of a 128-column block of the same 16 weight rows, so every weight byte is fetched
// from HBM once per 128 columns (the other seven waves hit the CU's L1).  EPI 1 (SiLU(gate) * up) runs the gate tile's K
// loop, keeps its four results, then the up tile's.
#include "ps_gemv_dev.h"

namespace {

typedef int g4k_i32x4 __attribute__((ext_vector_type(4)));

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

// one super-block of one tile: wq[u] = this lane's weight dword (row l % 16, bytes 32 kb + 4u..), hA = the header of row
// l % 16, hD[r] = the headers of rows 4 kb + r, bq = the B fragments, yd / b16 = the column's scale and 16-sums
template <int EXP>
__device__ __forceinline__ void g4k_superblock(G4KAcc &T, const uint32_t (&wq)[8], const uint4 hA, const uint32_t (&hD)[4], const ps_u32x4 (&bq)[4],
                                               const float yd, const ps_u32x4 b16a, const ps_u32x4 b16b, const int kb) {
    // A operands: nibbles times the 3-bit halves of the sub-block scales 2 kb, 2 kb + 1 (get_scale_min_k4, branch-free:
    // sub-blocks 0..3 sit in the low 6 bits of scale bytes 0..3, sub-blocks 4..7 are spread over bytes 8..11 and the top
    // bits of bytes 0..3)
    const int is0 = 2 * kb, sh0 = 8 * (is0 & 3), sh1 = sh0 + 8;
    const uint32_t a0 = (hA.y >> sh0) & 0xff, a1 = (hA.y >> sh1) & 0xff, c0 = (hA.w >> sh0) & 0xff, c1 = (hA.w >> sh1) & 0xff;
    const int sc0 = kb < 2 ? (int)(a0 & 63) : (int)((c0 & 0xF) | ((a0 >> 6) << 4));
    const int sc1 = kb < 2 ? (int)(a1 & 63) : (int)((c1 & 0xF) | ((a1 >> 6) << 4));
    const uint32_t f0h = (uint32_t)(sc0 >> 3) * 0x00010001u, f0l = (uint32_t)(sc0 & 7) * 0x00010001u;
    const uint32_t f1h = (uint32_t)(sc1 >> 3) * 0x00010001u, f1l = (uint32_t)(sc1 & 7) * 0x00010001u;
    float dr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) dr[r] = __fmul_rn(yd, ps_h2f((uint16_t)(hD[r] & 0xffff)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pkmul = [](uint32_t a, uint32_t f) { u16x2 va, vf; __builtin_memcpy(&va, &a, 4); __builtin_memcpy(&vf, &f, 4); va = va * vf; uint32_t o; __builtin_memcpy(&o, &va, 4); return o; };
    // the MFMAs in rounds of four independent ones (the second round of a group takes the first round's results, shifted,
    // as its C input): the matrix core's latency is covered by the other three, not by wait states
#pragma unroll
    for (int uh = 0; uh < 8; uh += 4) { // (four at a time: eight keep 48 more registers alive than the 168 of a nine-wave workgroup allow)
        g4k_i32x4 cc[4];
        long al[4], bb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int u = uh + k;
            const uint32_t lo = wq[u] & 0x0F0F0F0Fu, hi = (wq[u] >> 4) & 0x0F0F0F0Fu;
            const long a_hi = (EXP & 8) ? g4k_pack(lo, hi) : g4k_pack(pkmul(lo, f0h), pkmul(hi, f1h));
            al[k] = (EXP & 8) ? g4k_pack(hi ^ f0l, lo ^ f1l) : g4k_pack(pkmul(lo, f0l), pkmul(hi, f1l));
            const uint32_t b0 = (u & 1) ? bq[u >> 1].z : bq[u >> 1].x, b1 = (u & 1) ? bq[u >> 1].w : bq[u >> 1].y;
            bb[k] = g4k_pack(b0, b1);
            const g4k_i32x4 z = {0, 0, 0, 0};
            if (EXP & 32) { cc[k] = z; cc[k][0] = (int)a_hi ^ (int)bb[k]; cc[k][1] = (int)(a_hi >> 32); }
            else cc[k] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a_hi, bb[k], z, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (EXP & 32) { cc[k] = cc[k] << 3; cc[k][2] ^= (int)al[k]; cc[k][3] ^= (int)(al[k] >> 32); }
            else cc[k] = __builtin_amdgcn_mfma_i32_16x16x32_i8(al[k], bb[k], cc[k] << 3, 0, 0, 0); // sumi[u] of rows 4 kb + r, this lane's column
        }
        if (EXP & 1) {
            const g4k_i32x4 x = (cc[0] ^ cc[1]) ^ (cc[2] ^ cc[3]);
#pragma unroll
            for (int r = 0; r < 4; r++) T.acc[r][uh] = __int_as_float(__float_as_int(T.acc[r][uh]) ^ x[r]);
        } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int r = 0; r < 4; r++) T.acc[r][uh + k] = __fmaf_rn(dr[r], (float)cc[k][r], T.acc[r][uh + k]);
        }
        }
    }
    if (!(EXP & 2)) {
        // acc_m lane v: prod = mins[2v] * q8s[2v] + mins[2v+1] * q8s[2v+1] -- a K = 4 contraction over the four 16-sums of
        // sub-blocks 2v, 2v + 1 (q8s = the sum of two).  Every factor is an integer fp16 holds exactly (mins <= 63,
        // |16-sum| <= 2032) and every partial sum is below 2^24, so v_mfma_f32_16x16x16_f16 returns (float)prod itself:
        // the lanes of k-group 0 supply A[row l % 16] = (m[2v], m[2v], m[2v+1], m[2v+1]), the others zeros; B is the
        // column's fp16 16-sums as the quantizer stored them.
        typedef _Float16 g4k_h2 __attribute__((ext_vector_type(2)));
        typedef _Float16 g4k_h4 __attribute__((ext_vector_type(4)));
        typedef float g4k_f4 __attribute__((ext_vector_type(4)));
        const uint32_t lm = kb == 0 ? 0xffffffffu : 0u;
        const uint32_t mn03 = hA.z & 0x3f3f3f3fu & lm;
        const uint32_t mn47 = (((hA.w >> 4) & 0x0f0f0f0fu) | (((hA.z >> 6) & 0x03030303u) << 4)) & lm;
        float dmin[4];
#pragma unroll
        for (int r = 0; r < 4; r++) dmin[r] = __fmul_rn(-yd, ps_h2f((uint16_t)(hD[r] >> 16)));
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
            const g4k_h4 am = {h0[0], h0[1], h1[0], h1[1]};
            const uint32_t bx = v == 0 ? b16a.x : v == 1 ? b16a.z : v == 2 ? b16b.x : b16b.z;
            const uint32_t by = v == 0 ? b16a.y : v == 1 ? b16a.w : v == 2 ? b16b.y : b16b.w;
            g4k_h2 g0, g1;
            __builtin_memcpy(&g0, &bx, 4); __builtin_memcpy(&g1, &by, 4);
            const g4k_h4 bm = {g0[0], g0[1], g1[0], g1[1]};
            const g4k_f4 zf = {0.f, 0.f, 0.f, 0.f};
            const g4k_f4 pr = __builtin_amdgcn_mfma_f32_16x16x16f16(am, bm, zf, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) T.accm[r][v] = __fmaf_rn(dmin[r], pr[r], T.accm[r][v]);
        }
    }
}

// (Measured and not kept, round 2: preparing the MFMA A operands and the decoded mins / scales ONCE per workgroup, by the
// staging threads, and parking those in LDS instead of the raw dwords halves the VALU work of a wave's step (375 -> ~185
// instructions) but costs registers the nine-wave workgroup does not have (spills whose reloads queue behind the prefetch
// loads: 5.3 k tok/s) and, with the spills removed for the experiment by dropping the mins chains, gains only 12 %
// (7.95 k vs 7.1 k tok/s): the step is not VALU-bound alone.)
// All eight computing waves of a workgroup walk the same 16 weight rows.  Two things make a straight version (every wave
// loading its own operands; measured, removed) slow: the weights come from HBM (~2 us per super-block with nothing
// but the next loads to hide behind), and the CU's address unit: a wave's eight dword loads of A touch 16 cache lines each,
// eight waves repeat them, and the column metadata adds 48 more line look-ups per wave -- ~2200 cycles of address
// processing per super-block step against ~800 of arithmetic.  So:
//   * a NINTH wave touches the 18 lines of a super-block (2 KiB of quants + 2 x 128 B of headers; one dword per 128-B
//     line) G4K_LEAD steps ahead: the weights are in the L2 when they are asked for;
//   * the 512 computing threads fetch ONE dword each of the next super-block (coalesced, one step ahead) and park it in an
//     LDS stage (rows padded to 144 B: a wave's A reads are at most 2-way bank conflicts); everybody reads A operands and
//     headers from LDS behind ONE barrier per step (three stages: the one being written is never one a slow wave may
//     still read).
constexpr int G4K_LEAD = 10, G4K_ROW = 144, G4K_STAGE = 16 * G4K_ROW + 256;
// stages 0 .. nsb-1: tile of (qs0, aux0); stages nsb .. n_stages-1 (EPI 1): the same tile of (qs1, aux1)
template <int EXP>
__device__ __forceinline__ void g4k_warm_wave(const uint8_t *qs0, const uint8_t *aux0, const uint8_t *qs1, const uint8_t *aux1, const int tile,
                                              const int nsb, const int n_stages, float *never) {
    const int lane = threadIdx.x & 63;
    const size_t off  = lane < 16 ? ((size_t)(2 * tile + (lane >> 3)) * nsb << 10) + (lane & 7) * 128 : (size_t)(2 * tile + (lane & 1)) * nsb * 128;
    const size_t step = lane < 16 ? 1024 : 128;
    const uint8_t *b0 = (lane < 16 ? qs0 : aux0) + off, *b1 = (lane < 16 ? qs1 : aux1) + off;
    auto touch = [&](int st) -> uint32_t {
        if (lane >= 18 || st >= n_stages) return 0u;
        return *(const uint32_t *)(st < nsb ? b0 + (size_t)st * step : b1 + (size_t)(st - nsb) * step);
    };
    uint32_t sink = 0;
    for (int st = 0; st < G4K_LEAD; st++) sink ^= touch(st);
    if (EXP & 16) { for (int st = G4K_LEAD; st < n_stages; st++) sink ^= touch(st); if (sink == 0x9e3779b9u && n_stages < 0) never[0] = 0.f; return; }
    __syncthreads(); // (stage 0 is parked)
    for (int s0 = 0; s0 < n_stages; s0 += 4) { // one barrier per step, as the computing waves; four touches in flight
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = touch(s0 + k + G4K_LEAD); __syncthreads(); }
#pragma unroll
        for (int k = 0; k < 4; k++) sink ^= v[k];
    }
    if (sink == 0x9e3779b9u && n_stages < 0) never[0] = 0.f; // (keeps the touches alive; never true)
}

// one tile of the staged walk: `s_first` = index of its first step in the workgroup's step sequence (LDS stage = step % 3)
template <int EXP>
__device__ __forceinline__ void g4k_tile_staged(const uint8_t *qs, const uint8_t *aux, const int tile, const int nsb, const int8_t *qf_ct,
                                                const uint8_t *mf_ct, const int mc, char *lds, const int s_first, const bool more,
                                                const uint8_t *qs_next, const uint8_t *aux_next, float (&y)[4]) {
    const int lane = threadIdx.x & 63, m = lane & 15, kb = lane >> 4, t = threadIdx.x;
    // this thread's dword of a super-block: unit t >> 8 (row group 2 * tile + (t >> 8)), dword t & 255 = [r][u][j]
    const size_t oq = ((size_t)(2 * tile + (t >> 8)) * nsb << 10) + (t & 255) * 4, oh = (size_t)(2 * tile + ((t >> 5) & 1)) * nsb * 128 + (t & 31) * 4;
    const int lq = ((t >> 8) * 8 + ((t & 255) >> 5)) * G4K_ROW + (t & 31) * 4, lh = 16 * G4K_ROW + (t & 63) * 4;
    G4KAcc T;
    T.clear();
    ps_u32x4 bn[4]; // the B fragments of the next super-block (L2: a few hundred cycles -- one step ahead is enough)
#pragma unroll
    for (int up = 0; up < 4; up++) bn[up] = *(const ps_u32x4 *)(qf_ct + up * 1024 + lane * 16);
    for (int sb = 0; sb < nsb; sb++) {
        // next step's weights: one dword per thread (the first super-block of the following tile after the last one)
        const bool last = sb + 1 == nsb;
        const uint8_t *nqs = last ? qs_next : qs, *nax = last ? aux_next : aux;
        const int nsbi = last ? 0 : sb + 1;
        uint32_t nq = 0, nh = 0;
        if (!last || more) {
            nq = *(const uint32_t *)(nqs + oq + ((size_t)nsbi << 10));
            if (t < 64) nh = *(const uint32_t *)(nax + oh + (size_t)nsbi * 128);
        }
        const char *st = lds + ((s_first + sb) % 3) * G4K_STAGE;
        uint32_t wq[8];
#pragma unroll
        for (int u = 0; u < 8; u++) wq[u] = *(const uint32_t *)(st + m * G4K_ROW + u * 16 + kb * 4);
        const uint4 hA = *(const uint4 *)(st + 16 * G4K_ROW + m * 16);
        uint32_t hD[4]; // (d, dmin) of the result rows 4 kb + r
#pragma unroll
        for (int r = 0; r < 4; r++) hD[r] = *(const uint32_t *)(st + 16 * G4K_ROW + (kb * 4 + r) * 16);
        const ps_u32x4 bq[4] = {bn[0], bn[1], bn[2], bn[3]};
        const uint8_t *mfs = mf_ct + (size_t)sb * 576; // the (column tile, super-block) block: d[16], then bsums[16][16]
        const float yd = *(const float *)(mfs + mc * 4);
        const ps_u32x4 b16a = *(const ps_u32x4 *)(mfs + 64 + mc * 32), b16b = *(const ps_u32x4 *)(mfs + 64 + mc * 32 + 16);
        if (!(EXP & 4)) {
            const int nb = last ? 0 : sb + 1; // (the following tile of an EPI 1 pair meets the same columns from super-block 0)
#pragma unroll
            for (int up = 0; up < 4; up++) bn[up] = *(const ps_u32x4 *)(qf_ct + ((size_t)nb << 12) + up * 1024 + lane * 16);
        }
        g4k_superblock<EXP>(T, wq, hA, hD, bq, yd, b16a, b16b, kb);
        if (!(EXP & 16) && (!last || more)) {
            char *sn = lds + ((s_first + sb + 1) % 3) * G4K_STAGE;
            *(uint32_t *)(sn + lq) = nq;
            if (t < 64) *(uint32_t *)(sn + lh) = nh;
        }
        if (!(EXP & 16)) __syncthreads();
    }
    T.reduce(y);
}

// nine waves: the eight column tiles of ONE row task (128 columns per workgroup) + the warm-up wave
template <int EPI, int EXP>
__global__ __launch_bounds__(576) void gemm4k_kernel(const G4KParams p) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    // wave -> (row task, column tile)
    const int ct = (int)blockIdx.y * 8 + (wave & 7);
    const int task = (int)blockIdx.x; // (grid.x = tasks exactly; every wave stays for the barriers)
    __shared__ __attribute__((aligned(16))) char lds[3 * G4K_STAGE];
    int wi = 0, tile = task;
    if (EPI != 1) {
        if (p.n_w > 1 && tile >= p.w[0].n_tiles) { tile -= p.w[0].n_tiles; wi = 1; }
        if (p.n_w > 2 && wi == 1 && tile >= p.w[1].n_tiles) { tile -= p.w[1].n_tiles; wi = 2; }
    }
    const G4KMat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
    const int col = ct * 16 + m, colc = col < p.bs ? col : p.bs - 1; // (the last tile may be ragged: clamp the column metadata)
    const int ctc = ct * 16 < p.bs ? ct : (p.bs - 1) / 16; // (a wave past the batch walks the last tile's columns and stores nothing)
    const int8_t *qf_ct = p.qf + ((size_t)ctc * p.nsb << 12);
    const uint8_t *mf_ct = p.mf + (size_t)ctc * p.nsb * 576;
    const int mc = colc & 15;
    float y[4];
    {
        if (wave == 8) { // the warm-up wave
            g4k_warm_wave<EXP>(W.qs, W.aux, EPI == 1 ? p.w[1].qs : W.qs, EPI == 1 ? p.w[1].aux : W.aux, tile, p.nsb, EPI == 1 ? 2 * p.nsb : p.nsb, W.out);
            return;
        }
        { // stage 0: the tile's first super-block
            const int t = threadIdx.x;
            const size_t oq = ((size_t)(2 * tile + (t >> 8)) * p.nsb << 10) + (t & 255) * 4, oh = (size_t)(2 * tile + ((t >> 5) & 1)) * p.nsb * 128 + (t & 31) * 4;
            *(uint32_t *)(lds + ((t >> 8) * 8 + ((t & 255) >> 5)) * G4K_ROW + (t & 31) * 4) = *(const uint32_t *)(W.qs + oq);
            if (t < 64) *(uint32_t *)(lds + 16 * G4K_ROW + t * 4) = *(const uint32_t *)(W.aux + oh);
        }
        __syncthreads();
        g4k_tile_staged<EXP>(W.qs, W.aux, tile, p.nsb, qf_ct, mf_ct, mc, lds, 0, EPI == 1, p.w[1].qs, p.w[1].aux, y);
        if (EPI == 1) {
            float yu[4];
            g4k_tile_staged<EXP>(p.w[1].qs, p.w[1].aux, tile, p.nsb, qf_ct, mf_ct, mc, lds, p.nsb, false, nullptr, nullptr, yu);
#pragma unroll
            for (int r = 0; r < 4; r++) y[r] = ps_silu_mul(y[r], yu[r]);
        }
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
    const int n_ct = (int)((bs + 15) / 16);
    // Below eight column tiles a workgroup's waves would not share their weight rows any more, and the kernels that spread
    // a row group's integer work over producer waves (gemm8) are ahead there: tree forward of the 8B shape, ms by width,
    // this kernel / gemm8: 2: 12.5 / 4.9, 12: 13.7 / 6.4, 32: 14.4 / 9.2, 64: 18.3 / 13.3, 128: 17.0 / 22.9
    // (profiles/r02_tree_forward_latency_8b.json).
    if (n_ct < 8 || p.nsb % 4) return -1;
    (void)n_cu;
    const dim3 grid((unsigned)p.n_tasks, (unsigned)((n_ct + 7) / 8));
    extern int g_g4_flags;
#define G4K_L(E) case E: if (epi == 1) hipLaunchKernelGGL((gemm4k_kernel<1, E>), grid, dim3(576), 0, st, p); else hipLaunchKernelGGL((gemm4k_kernel<0, E>), grid, dim3(576), 0, st, p); break;
    switch (g_g4_flags) { G4K_L(1) G4K_L(2) G4K_L(3) G4K_L(4) G4K_L(8) G4K_L(16) G4K_L(32) G4K_L(11) G4K_L(15) G4K_L(31) G4K_L(63) default: G4K_L(0) }
    return 0;
}
