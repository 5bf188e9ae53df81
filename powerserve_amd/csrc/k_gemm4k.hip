// Q4_K batched mat-mul (prefill chunks, tree verify) on v_mfma_f32_16x16x32_f16 -- bit-exact with ggml_vec_dot_q4_K_q8_K.
//
// The AVX2 kernel (libs/ggml/src/ggml-quants.c:7809-7873) keeps, per super-block of 256 weights, ONE int32 per accumulator
// lane u:  sumi[u] = sum over the 8 sub-blocks g of  scale[g] * dot4(q[g][4u..4u+3], y[g][4u..4u+3]),  accumulated in
// int32 before the single  acc[u] = fma(d * y.d, (float)sumi[u], acc[u]).  Integers associate, and the 32 elements lane u
// owns in a super-block are exactly one K = 32 contraction: for a tile of 16 weight rows x 16 activation columns and one u,
//     D[row][col] = sum_k A[row][k] * B[k][col],   k = (g, e),  A = q4 * scale[g],  B = y.
// Every factor is an integer fp16 holds exactly (q4 * scale <= 15 * 63 = 945 < 2048, |y| <= 127), every product is exact in
// fp32 and every partial sum stays below 2^24 (32 * 945 * 127 < 3.9 M): v_mfma_f32_16x16x32_f16 returns (float)sumi[u]
// ITSELF, whatever order it adds in -- no scale split, no shift, no int-to-float conversion.  The mins term
// (ggml-quants.c:7831-7834) per accumulator lane v,  prod[v] = mins[2v] * q8sum[2v] + mins[2v+1] * q8sum[2v+1],  is a K = 4
// contraction over the four 16-sums of the two sub-blocks (mins <= 63, |16-sum| <= 2032) on v_mfma_f32_16x16x16_f16 the same
// way.  Everything after sumi / prod is the reference's fp32 arithmetic per (row, column, lane): the eight acc chains, the
// four acc_m chains, hsum_float_8.
//
// Operand layout: lane l supplies A[i = l % 16][k = 8 * (l / 16) ..+7] and B[k = 8 * (l / 16) ..+7][j = l % 16] (8 fp16 =
// 16 B each); result register r of lane l is D[i = 4 * (l / 16) + r][j = l % 16].  With kb = l / 16 the lane's eight k are
// the elements 4u + (0, 2, 1, 3) of sub-block 2 kb, then of sub-block 2 kb + 1 (the order the nibbles fall out in):
//   * A: the lane's 8 weights come from ONE dword of the lane-major weight layout (ps_internal.h): byte 32 kb + 4u + e of
//     the super-block holds element e of sub-block 2 kb in its low nibble and of sub-block 2 kb + 1 in its high nibble;
//   * B: the quantizer writes a second, fragment-major fp16 copy of the Q8_K quants (ps_act::qf): per (16 columns,
//     super-block) 8 KiB laid out [u][lane][16 B], so that a wave's B operands for one u are one fully coalesced 1 KiB
//     load; the column metadata rides tile-major next to it (ps_act::mf).
//
// A workgroup is twelve waves on TWO 16-row tiles x 64 columns (an EPI 1 pair: the gate tile and the up tile of the same
// rows; else 32 consecutive rows of one matrix).  Waves 0-7 COMPUTE: wave = (column tile ct = w % 4, accumulator half
// uh = w / 4) owns, for BOTH row tiles, the accumulator lanes u = 4 uh .. 4 uh + 3 and the mins lanes v = 2 uh, 2 uh + 1 of
// one 16 x 16 tile (48 fp32 chains per lane): a B operand fetched from L2 meets two row tiles, which halves the L2 traffic
// per MFMA -- what bounded the one-row-tile form (17 TB/s of fragment reads, tools/gpu_g4k_timeline.py) -- without the 96
// chain registers of two whole tiles.  The two halves of a tile meet once, at the end, through LDS (hsum_float_8's first
// level adds lane u to lane u + 4: exactly the two halves).  Waves 8-11 PRODUCE: everything the computing waves would
// otherwise each derive from the same 32 rows -- the nibbles times their scales as ready fp16 A operands, the mins as
// fp16 A operands, d and dmin as fp32 -- is made ONCE per super-block (producer h: accumulator lanes 2h, 2h + 1 of all 32
// rows; lane = (row, k-group pair)) and parked in LDS ahead of the consumers, one barrier per pair of steps.  The producers
// own their memory pipeline: a register ring of G4K_RING super-blocks is in flight from HBM, nobody waits on a weight load.
// (Round-2 history, tools/g4k_exp.py and tools/gpu_g4k_timeline.py: a per-wave int8 version -- scale split 8 hi + lo, two
// v_mfma_i32_16x16x32_i8 with a shift between -- spent a third of its vector instructions on the shared derivations and,
// once those had moved to producers, 64 of its remaining 176 instructions per step on the shifts and conversions the fp16
// form does not have.)
#include "ps_gemv_dev.h"

namespace {

typedef _Float16 g4k_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 g4k_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 g4k_h4 __attribute__((ext_vector_type(4)));
typedef float g4k_f4 __attribute__((ext_vector_type(4)));

struct G4KMat {
    const uint8_t *qs, *aux; // Q4_K: row-group units + headers;  Q5_K: per-row planes qs [N][nsb][u][16 B], aux = headers [N][nsb] 16 B
    const uint8_t *qh;       // Q5_K: the fifth bits, [N][nsb][u][4 B]
    float *out;
    const float *bias;
    int64_t N, ldo;
    int n_tiles; // N / 16
};
struct G4KParams {
    G4KMat w[3];
    int wt;                  // PS_Q4_K or PS_Q5_K (one producer each; the consumers are the same)
    int n_w, nsb, bs, n_tasks, n_cb, n_items; // n_cb: 64-column blocks; items = (task, column block), tasks padded to a multiple of 8
    int cbx;                 // > 0: column blocks per XCD of the XCD-aware item order (g4k_item); 0: the round-2 order
    const float *residual;
    const _Float16 *qf; // fragment-major fp16 quants
    const uint8_t *mf;  // tile-major column metadata (ps_act::mf)
    unsigned long long *dbg; // timeline slots (ps_hip_debug_timeline keys 48..50, 52), or null
    psk_rope_kv rope;        // rope_on (Q / K / V launches, adjacent-pair RoPE): the epilogue rotates Q and K and appends K, V to the caches
    int rope_on;
};

// One LDS stage = one super-block of the workgroup's 32 rows (row tile t = row / 16), as the consumers want it:
//   [kb][row][u] 16 B = the fp16 A operand of lane (row % 16, kb) for accumulator lane u; rows of 8 operands padded to 144 B
//                       (9 bank quads: the 16 rows of a tile land on 16 distinct quads) and the kb planes a multiple of 16
//                       quads apart, so each 16-lane group of a ds_read_b128 (which mixes lanes of two kb) is conflict-free
//   [row] 32 B   = the mins as the four fp16 A operands (m[2v], m[2v], m[2v+1], m[2v+1]), v = 0..3
//   [row] 8 B    = (d, dmin) as fp32
// (G4K_PAD: the operand planes of k-groups 2, 3 sit 16 B further on than those of k-groups 0, 1.  A plane is 4608 B = 18 x 256 B, so
// without a shift the two lanes of a producer row -- k-groups (e, 2 + e), the same row offset -- store to the SAME banks at different
// addresses: every ds_write_b128 of the producers takes two passes (SQ_LDS_BANK_CONFLICT = 28 % of SQ_LDS_IDX_ACTIVE,
// profiles/r03_pmc_sq_prefill.txt).  A consumer read never mixes k-groups {0, 1} with {2, 3} in one 16-lane bank group, so its
// conflict-free row stride survives any such shift.  tools/lds_bank_check.py, written after the round's last GPU minute, says that
// under the guide's STORE bank function (32 banks, groups of 8 lanes) 16 B only moves the collision to the neighbouring row and that
// 64 B removes it.  Round 4 timed it (tools/ab_build.py, profiles/r04_prefill_ab.txt): gate/up launch of a 512-column sequence 378.5 -> 375.6 us,
// prefill 17.33 k -> 17.40 k tok/s warm -- the store conflicts were never what bounds the kernel; 64 stays because it is the conflict-free value.)
constexpr int G4K_PAD = 64;
constexpr int G4K_RS = 144, G4K_KB = 32 * G4K_RS, G4K_MINS = 4 * G4K_KB + G4K_PAD, G4K_DD = G4K_MINS + 32 * 32, G4K_STAGE = G4K_DD + 32 * 8;
__device__ __forceinline__ int g4k_plane(const int kb) { return kb * G4K_KB + (kb >> 1) * G4K_PAD; } // byte offset of k-group kb's operand plane in a stage
constexpr int G4K_NC = 8, G4K_NP = 4, G4K_RING = 4; // computing waves, producing waves, super-blocks in flight per producer
constexpr int G4K_NST = 4;                          // LDS stages
constexpr int G4K_XCH = G4K_NST * G4K_STAGE + 32;   // after the stages and the 32 zero bytes: the end-of-tile exchange, [ct][lane][48 floats]
constexpr int G4K_TAB = G4K_XCH + 4 * 64 * 48 * 4;   // glibc expf's exp2 table (ps_expf.h), for the SiLU epilogue: an LDS look-up, not a global one
constexpr int G4K_LDS = G4K_TAB + PS_EXP2F_N * 8;

// the two row tiles of a task
struct G4KRows { const uint8_t *qs[2], *aux[2], *qh[2]; int tile[2]; };

// ---- producers.  Wave h (0..3) makes the accumulator lanes u = 2h, 2h + 1 of all 32 rows; lane = (row l / 2, kb pair p = l % 2):
// its weight dwords [row][u][kb = 2p, 2p + 1] are one 8-B load per u.
// Q5_K (ggml_vec_dot_q5_K_q8_K, ggml-quants.c:8237-8320: the Q4_K kernel with a fifth bit per weight, value <= 31, x scale
// <= 1953 < 2048: still one exact fp16 operand): hq = the lane's four qh bytes (bit g of byte i = fifth bit of element 4u + i of
// sub-block g); zero for Q4_K.
template <int UPP, int WT> // accumulator lanes per producer wave: u = UPP hw .. UPP hw + UPP - 1
__device__ __forceinline__ void g4k_produce(const uint2 (&qq)[UPP], const uint32_t (&hq)[UPP], const uint4 h, char *st, const int row, const int hw, const int p) {
    // the four sub-block scales 4p .. 4p+3 (get_scale_min_k4: 0..3 sit in the low 6 bits of scale bytes 0..3, 4..7 are
    // spread over bytes 8..11 and the top bits of bytes 0..3) as fp16 pairs (s, s) and (-1024 s, -1024 s)
    const uint32_t scb = p ? ((h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4)) : (h.y & 0x3f3f3f3fu);
    const g4k_h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f}, km1024 = {(_Float16)-1024.f, (_Float16)-1024.f};
    g4k_h2 sc[4], nc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { // bytes (s, 0x64, s, 0x64) = the fp16 pair (1024 + s, 1024 + s); minus 1024 is exact
        const uint32_t t = __builtin_amdgcn_perm(0x64646464u, scb, 0x04000400u | (uint32_t)(i * 0x00010001u));
        g4k_h2 v;
        __builtin_memcpy(&v, &t, 4);
        sc[i] = v - k1024;
        nc[i] = sc[i] * km1024; // (<= 64512: exact)
    }
#pragma unroll
    for (int j = 0; j < UPP; j++) { // u = UPP hw + j
        const uint2 q = qq[j];
#pragma unroll
        for (int e = 0; e < 2; e++) { // kb = 2p + e: sub-blocks 2 kb (low nibbles), 2 kb + 1 (high nibbles) = scales 2e, 2e + 1 of the four
            const uint32_t w = e ? q.y : q.x;
            // nibble pairs as fp16 (1024 + n): (e0, e2), (e1, e3) of sub-block 2 kb, then of 2 kb + 1;  fma(1024 + n, s, -1024 s) = n s, exact
            uint32_t t[4] = {(w & 0x000F000Fu) | 0x64006400u, ((w >> 8) & 0x000F000Fu) | 0x64006400u,
                             ((w >> 4) & 0x000F000Fu) | 0x64006400u, ((w >> 12) & 0x000F000Fu) | 0x64006400u};
            if (WT == PS_Q5_K) { // the fifth bits of sub-blocks g0 = 4p + 2e (t[0], t[1]) and g0 + 1 (t[2], t[3])
                const uint32_t hg = hq[j] >> (4 * p + 2 * e);
                t[0] |= (hg & 0x00010001u) << 4; t[1] |= ((hg >> 8) & 0x00010001u) << 4;
                t[2] |= ((hg >> 1) & 0x00010001u) << 4; t[3] |= ((hg >> 9) & 0x00010001u) << 4;
            }
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                g4k_h2 v;
                __builtin_memcpy(&v, &t[k], 4);
                v = __builtin_elementwise_fma(v, sc[2 * e + (k >> 1)], nc[2 * e + (k >> 1)]);
                __builtin_memcpy(&o[k], &v, 4);
            }
            *(uint4 *)(st + g4k_plane(2 * p + e) + row * G4K_RS + (UPP * hw + j) * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    if (hw == 0 && p == 0) { // once per row: the mins operands
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
            h0 = h0 - k1024; h1 = h1 - k1024;
            __builtin_memcpy(&o[2 * v], &h0, 4); __builtin_memcpy(&o[2 * v + 1], &h1, 4);
        }
        *(uint4 *)(st + G4K_MINS + row * 32) = make_uint4(o[0], o[1], o[2], o[3]);
        *(uint4 *)(st + G4K_MINS + row * 32 + 16) = make_uint4(o[4], o[5], o[6], o[7]);
    }
    if (hw == 1 && p == 0) *(float2 *)(st + G4K_DD + row * 8) = make_float2(ps_h2f((uint16_t)(h.x & 0xffff)), ps_h2f((uint16_t)(h.x >> 16))); // (d, dmin)
}

// Two adjacent result rows (row0 even) of column `col` of matrix wi.  Plain: out[col][row0 .. row0 + 1].  With rope_on the
// launch is Q / K / V and the pair is exactly one RoPE pair (rope_append_kernel's arithmetic, k_attn.hip; ggml.c:15344-15358):
// Q is rotated into out, K rotated into its cache row (slot pos0 + col), V appended transposed -- the separate launch between the
// mat-mul and the attention disappears.
template <int ROPE> // 0: never, 1: always, 2: by p.rope_on
__device__ __forceinline__ void g4k_store_pair(const G4KParams &p, const int wi, const G4KMat &W, const int col, const int64_t row0, const float x0, const float x1) {
    if (ROPE == 0 || (ROPE == 2 && !p.rope_on)) { *(float2 *)(W.out + (int64_t)col * W.ldo + row0) = make_float2(x0, x1); return; }
    const psk_rope_kv &R = p.rope;
    const int pos = R.state->pos0 + col; // cache slot
    if (wi == 2) {
        R.v_cache[row0 * R.n_ctx + pos] = x0; R.v_cache[(row0 + 1) * R.n_ctx + pos] = x1;
        if (R.v16) { R.v16[(int64_t)pos * R.kv_dim + row0] = (_Float16)x0; R.v16[(int64_t)pos * R.kv_dim + row0 + 1] = (_Float16)x1; }
        return;
    }
    const int rp = R.rope_pos ? R.rope_pos[col] : pos, i0 = (int)(row0 % R.head_size);
    float ra = x0, rb = x1;
    if (i0 < R.n_dims) {
        const float2 cs = *(const float2 *)(R.rope_table + (int64_t)rp * R.head_size + i0);
        ps_rope_pair(x0, x1, cs.x, cs.y, ra, rb);
    }
    if (wi == 0) *(float2 *)(W.out + (int64_t)col * W.ldo + row0) = make_float2(ra, rb);
    else {
        *(float2 *)(R.k_cache + (int64_t)pos * R.kv_dim + row0) = make_float2(ra, rb);
        if (R.k16) { R.k16[(int64_t)pos * R.kv_dim + row0] = (_Float16)ra; R.k16[(int64_t)pos * R.kv_dim + row0 + 1] = (_Float16)rb; }
    }
}

// item i -> (task, column block).
// cbx == 0 (round 2): 16 consecutive items are 8 tasks x 2 column blocks (for n_cb = 2), the two blocks of a task 8 apart -- the same XCD
// at the same time, so the second one finds the weights in that L2.  With n_cb = 8 (a 512-column sequence) that puts ALL eight column blocks
// on every XCD: their fragment-major activations are 4.4 MB at K = 4096, more than the XCD's 4 MB L2, cycled once per item -- the L2 thrashes
// and the launch fetches 16 x its weights from the fabric (profiles/r04_pmc_traffic.json: 1070 MB for 66 MB).
// cbx > 0 (round 5; needs gridDim.x == 256, n_items % 256 == 0, n_cb a power of two, cbx | n_cb): workgroup b sits on XCD b % 8 (slot b / 8
// of its 32).  The n_cb / cbx groups of cbx column blocks are dealt over the XCDs; the 8 cbx / n_cb XCDs that share a group split the tasks
// between them.  An XCD then cycles only cbx x 0.55 MB of activations and streams 1 / (8 cbx / n_cb) of the weights, each weight tile met by
// cbx workgroups of that XCD at the same time.
__device__ __forceinline__ void g4k_item(const G4KParams &p, const int i, int &task, int &cb) {
    if (p.cbx) {
        const int b = i & 255, k = i >> 8, xcd = b & 7, slot = b >> 3;
        const int G = p.n_cb / p.cbx, TC = 8 / G, NT = 32 / p.cbx;
        cb = (xcd % G) * p.cbx + slot % p.cbx;
        task = (k * NT + slot / p.cbx) * TC + xcd / G;
        return;
    }
    cb = (i >> 3) % p.n_cb;
    task = (i / (8 * p.n_cb)) * 8 + (i & 7);
}
template <int EPI>
__device__ __forceinline__ G4KRows g4k_rows(const G4KParams &p, const int task, int &wi, int &pair) {
    wi = 0; pair = task;
    if (EPI != 1) {
        if (p.n_w > 1 && pair >= p.w[0].n_tiles / 2) { pair -= p.w[0].n_tiles / 2; wi = 1; }
        if (p.n_w > 2 && wi == 1 && pair >= p.w[1].n_tiles / 2) { pair -= p.w[1].n_tiles / 2; wi = 2; }
    }
    const G4KMat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
    G4KRows R;
    if (EPI == 1) { R.qs[0] = p.w[0].qs; R.aux[0] = p.w[0].aux; R.qh[0] = p.w[0].qh; R.qs[1] = p.w[1].qs; R.aux[1] = p.w[1].aux; R.qh[1] = p.w[1].qh; R.tile[0] = R.tile[1] = task; }
    else { R.qs[0] = R.qs[1] = W.qs; R.aux[0] = R.aux[1] = W.aux; R.qh[0] = R.qh[1] = W.qh; R.tile[0] = 2 * pair; R.tile[1] = 2 * pair + 1; }
    return R;
}
__device__ __forceinline__ int g4k_next_item(const G4KParams &p, int i) { // the next item of this workgroup with a real task, or n_items
    for (i += (int)gridDim.x; i < p.n_items; i += (int)gridDim.x) {
        int t, c;
        g4k_item(p, i, t, c);
        if (t < p.n_tasks) break;
    }
    return i < p.n_items ? i : p.n_items;
}

// The producers of a PERSISTENT workgroup: the stage stream runs on across the workgroup's items -- the ring is already
// loading the next item's first super-blocks while the consumers finish this one, and its first two stages are parked
// before the consumers' end-of-item exchange barrier (X), so a new item starts at full speed.
template <int EPI, int NPW, int RING, int WT> // NPW producer waves: wave hw makes the accumulator lanes 8 / NPW * hw ..; RING super-blocks in flight (nsb % RING == 0)
__device__ __forceinline__ void g4k_producer_wave(const G4KParams &p, int item, char *lds, const int hw, unsigned long long *dbg) {
    constexpr int UPP = 8 / NPW;
    const int lane = threadIdx.x & 63, nsb = p.nsb;
    int dbg_n = 1;
    auto mark = [&](int g) { if (dbg && (g < 8 || (g & 7) == 7) && dbg_n < 29) dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); };
    const int row = lane >> 1, pp = lane & 1, rt = row >> 4, unit = (row >> 3) & 1, r8 = row & 7;
    const uint8_t *qb, *hb, *fb = nullptr; // the load cursor's item (fb: Q5_K's fifth bits)
    auto point = [&](int it) {
        int task, cb, wi, pair;
        g4k_item(p, it, task, cb);
        const G4KRows R = g4k_rows<EPI>(p, task, wi, pair);
        if (WT == PS_Q5_K) { // per-row planes
            const size_t gr = (size_t)(rt ? R.tile[1] : R.tile[0]) * 16 + (row & 15);
            qb = (rt ? R.qs[1] : R.qs[0]) + (gr * nsb * 8 + UPP * hw) * 16 + pp * 8;
            fb = (rt ? R.qh[1] : R.qh[0]) + (gr * nsb * 8 + UPP * hw) * 4;
            hb = (rt ? R.aux[1] : R.aux[0]) + gr * nsb * 16;
        } else {
            qb = (rt ? R.qs[1] : R.qs[0]) + ((size_t)(2 * (rt ? R.tile[1] : R.tile[0]) + unit) * nsb << 10) + (size_t)(r8 * 32 + UPP * hw * 4 + 2 * pp) * 4;
            hb = (rt ? R.aux[1] : R.aux[0]) + (size_t)(2 * (rt ? R.tile[1] : R.tile[0]) + unit) * nsb * 128 + (size_t)r8 * 16;
        }
    };
    uint2 rq[RING][UPP];
    uint32_t rf[RING][UPP];
    uint4 rh[RING];
    auto load = [&](int g, uint2 (&a)[UPP], uint32_t (&f)[UPP], uint4 &h) {
#pragma unroll
        for (int j = 0; j < UPP; j++) {
            if (WT == PS_Q5_K) { a[j] = *(const uint2 *)(qb + (size_t)g * 128 + j * 16); f[j] = *(const uint32_t *)(fb + (size_t)g * 32 + j * 4); }
            else { a[j] = *(const uint2 *)(qb + ((size_t)g << 10) + j * 16); f[j] = 0u; }
        }
        h = *(const uint4 *)(hb + (size_t)g * (WT == PS_Q5_K ? 16 : 128));
    };
    point(item);
#pragma unroll
    for (int k = 0; k < RING; k++) load(k, rq[k], rf[k], rh[k]);
    int c_item = item, c_g = RING; // the cursor: the ring's next loads
    bool first = true;
    while (item < p.n_items) {
        for (int g0 = 0; g0 < nsb; g0 += RING) { // (nsb % RING == 0: psk_gemm4k checks)
            if (c_g == nsb) { // the ring moves on to the workgroup's next item (past the last one: reloads that one's tail, unused)
                const int nx = g4k_next_item(p, c_item);
                if (nx < p.n_items) { c_item = nx; c_g = 0; point(nx); } else c_g = nsb - RING;
            }
#pragma unroll
            for (int k = 0; k < RING; k++) {
                uint2 a[UPP];
                uint32_t f[UPP];
#pragma unroll
                for (int j = 0; j < UPP; j++) { a[j] = rq[k][j]; f[j] = rf[k][j]; }
                const uint4 h = rh[k];
                load(c_g + k, rq[k], rf[k], rh[k]);
                g4k_produce<UPP, WT>(a, f, h, lds + ((g0 + k) & (G4K_NST - 1)) * G4K_STAGE, row, hw, pp);
                if (k & 1) {
                    if (g0 == 0 && k == 1 && !first) __syncthreads(); // X of the previous item: its consumers have exchanged
                    __syncthreads(); // one barrier per PAIR of stages: stages g0 + k - 1, g0 + k are parked
                }
                mark(g0 + k);
            }
            c_g += RING;
        }
        first = false;
        item = g4k_next_item(p, item);
    }
    __syncthreads(); // X of the last item
}

// ---- consumers
struct G4KMeta { float yd; ps_u32x4 b16; }; // the column's scale and the fp16 16-sums of sub-blocks 4 uh .. 4 uh + 3 of one super-block
__device__ __forceinline__ G4KMeta g4k_meta(const uint8_t *mf_ct, const int sb, const int mc, const int boff) { // boff: which 16 B of the column's 16-sums
    const uint8_t *mfs = mf_ct + (size_t)sb * 576; // the (column tile, super-block) block: d[16], then the fp16 16-sums [16][16]
    G4KMeta M;
    M.yd = *(const float *)(mfs + mc * 4);
    M.b16 = *(const ps_u32x4 *)(mfs + 64 + mc * 32 + boff);
    return M;
}

// the fp32 chains this wave keeps: for both row tiles, rows 4 kb + r, accumulator lanes 4 uh + k and mins lanes 2 uh + vv
// (kept as PAIRS OVER ROWS: a matrix instruction returns rows 4 kb .. 4 kb + 3 of one accumulator lane in four consecutive registers,
// so (row r, row r + 1) of lane k is a register pair as it comes and one v_pk_fma_f32 advances two chains with no copies.  As scalars
// the compiler paired lanes k, k + 1 of one row instead -- results of two different matrix instructions -- and spent two v_mov per
// packed fma putting them side by side: 28 of a step's 145 instructions.)
typedef float g4k_f2 __attribute__((ext_vector_type(2)));
struct G4KAcc {
    g4k_f2 acc[2][4][2], accm[2][2][2]; // acc[t][k][r >> 1][r & 1], accm[t][vv][r >> 1][r & 1]
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int k = 0; k < 4; k++) acc[t][k][h] = g4k_f2{0.f, 0.f};
                accm[t][0][h] = accm[t][1][h] = g4k_f2{0.f, 0.f};
            }
    }
    __device__ __forceinline__ float a(int t, int r, int k) const { return acc[t][k][r >> 1][r & 1]; }
    __device__ __forceinline__ float am(int t, int r, int vv) const { return accm[t][vv][r >> 1][r & 1]; }
};

// one super-block.  B[k] = this lane's B operand for accumulator lane 4 uh + k (loaded a step ago); as soon as both row
// tiles have met it its registers take the load for the NEXT super-block (nq)
template <int WT>
__device__ __forceinline__ void g4k_superblock(G4KAcc &T, const char *st, const char *zero, ps_u32x4 (&B)[4], const char *nq, const G4KMeta M,
                                               const int m, const int kb, const int uh) {
    const g4k_f4 zf = {0.f, 0.f, 0.f, 0.f};
    g4k_f2 dr[2][2], dmin[2][2]; // [t][r >> 1] = (row r, row r + 1)
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const g4k_f4 dda = *(const g4k_f4 *)(st + G4K_DD + (16 * t + 4 * kb) * 8), ddb = *(const g4k_f4 *)(st + G4K_DD + (16 * t + 4 * kb) * 8 + 16); // (d, dmin) of rows 4 kb + r
        dr[t][0] = g4k_f2{__fmul_rn(M.yd, dda[0]), __fmul_rn(M.yd, dda[2])}; dr[t][1] = g4k_f2{__fmul_rn(M.yd, ddb[0]), __fmul_rn(M.yd, ddb[2])};
        dmin[t][0] = g4k_f2{__fmul_rn(-M.yd, dda[1]), __fmul_rn(-M.yd, dda[3])}; dmin[t][1] = g4k_f2{__fmul_rn(-M.yd, ddb[1]), __fmul_rn(-M.yd, ddb[3])};
    }
    const char *ap = st + g4k_plane(kb) + m * G4K_RS + (4 * uh) * 16;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        g4k_h8 bv;
        __builtin_memcpy(&bv, &B[k], 16);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const ps_u32x4 ao = *(const ps_u32x4 *)(ap + t * 16 * G4K_RS + k * 16);
            g4k_h8 av;
            __builtin_memcpy(&av, &ao, 16);
            const g4k_f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, zf, 0, 0, 0); // (float)sumi[4 uh + k] of rows 4 kb + r, this lane's column
            T.acc[t][k][0] = __builtin_elementwise_fma(dr[t][0], __builtin_shufflevector(si, si, 0, 1), T.acc[t][k][0]); // acc = fma(d, (float)sumi, acc), rows r, r + 1
            T.acc[t][k][1] = __builtin_elementwise_fma(dr[t][1], __builtin_shufflevector(si, si, 2, 3), T.acc[t][k][1]);
        }
        B[k] = *(const ps_u32x4 *)(nq + k * 1024);
    }
    if (WT == PS_Q5_K) {
        // Q5_K keeps the mins in ONE scalar chain per (row, column), multiply then add (ggml-quants.c:8411, two roundings):
        // summs += dmin * (float)(sum over all eight sub-blocks of mins * q8sum) -- a K = 16 contraction: k-group kb supplies
        // mins lane v = kb's operand and the column's 16-sums 4 kb .. 4 kb + 3.  Half 0 keeps the chain (accm[.][.][0]).
        if (uh == 0) {
            const uint32_t bx = (kb & 1) ? M.b16.z : M.b16.x, by = (kb & 1) ? M.b16.w : M.b16.y; // (b16 = the 16 B holding lane kb's sums)
            g4k_h2 g0, g1;
            __builtin_memcpy(&g0, &bx, 4); __builtin_memcpy(&g1, &by, 4);
            const g4k_h4 bm = {g0[0], g0[1], g1[0], g1[1]};
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const uint2 ma = *(const uint2 *)(st + G4K_MINS + (16 * t + m) * 32 + kb * 8);
                g4k_h2 a0, a1;
                __builtin_memcpy(&a0, &ma.x, 4); __builtin_memcpy(&a1, &ma.y, 4);
                const g4k_h4 am = {a0[0], a0[1], a1[0], a1[1]};
                const g4k_f4 pr = __builtin_amdgcn_mfma_f32_16x16x16f16(am, bm, zf, 0, 0, 0);
                T.accm[t][0][0] = T.accm[t][0][0] + dmin[t][0] * __builtin_shufflevector(pr, pr, 0, 1); // multiply, then add: two roundings (-ffp-contract=off)
                T.accm[t][0][1] = T.accm[t][0][1] + dmin[t][1] * __builtin_shufflevector(pr, pr, 2, 3);
            }
        }
        return;
    }
    // acc_m lanes 2 uh, 2 uh + 1: the lanes of k-group 0 supply the row's mins operands, the others zeros; B = the column's fp16 16-sums
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const char *mp = kb == 0 ? st + G4K_MINS + (16 * t + m) * 32 + uh * 16 : zero;
        const uint4 ma = *(const uint4 *)mp;
#pragma unroll
        for (int vv = 0; vv < 2; vv++) {
            const uint32_t ax = vv ? ma.z : ma.x, ay = vv ? ma.w : ma.y, bx = vv ? M.b16.z : M.b16.x, by = vv ? M.b16.w : M.b16.y;
            g4k_h2 a0, a1, g0, g1;
            __builtin_memcpy(&a0, &ax, 4); __builtin_memcpy(&a1, &ay, 4); __builtin_memcpy(&g0, &bx, 4); __builtin_memcpy(&g1, &by, 4);
            const g4k_h4 am = {a0[0], a0[1], a1[0], a1[1]}, bm = {g0[0], g0[1], g1[0], g1[1]};
            const g4k_f4 pr = __builtin_amdgcn_mfma_f32_16x16x16f16(am, bm, zf, 0, 0, 0);
            T.accm[t][vv][0] = __builtin_elementwise_fma(dmin[t][0], __builtin_shufflevector(pr, pr, 0, 1), T.accm[t][vv][0]);
            T.accm[t][vv][1] = __builtin_elementwise_fma(dmin[t][1], __builtin_shufflevector(pr, pr, 2, 3), T.accm[t][vv][1]);
        }
    }
}

// twelve waves: (4 column tiles x 2 accumulator halves) on two row tiles + the four producers; persistent over the items
// blockIdx.x, blockIdx.x + gridDim.x, ... (the launcher keeps the column block of a workgroup fixed across its items)
template <int EPI, int WT>
__global__ __launch_bounds__((G4K_NC + G4K_NP) * 64) void gemm4k_kernel(const G4KParams p) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (threadIdx.x < 8) ((uint32_t *)(lds + G4K_NST * G4K_STAGE))[threadIdx.x] = 0u; // the zero operands (visible after the first barrier)
    if (EPI == 1 && threadIdx.x >= 64 && threadIdx.x < 64 + PS_EXP2F_N) ((uint64_t *)(lds + G4K_TAB))[threadIdx.x - 64] = ps_exp2f_tab[threadIdx.x - 64];
    int item = (int)blockIdx.x;
    {
        int t, c;
        g4k_item(p, item, t, c);
        if (t >= p.n_tasks) item = g4k_next_item(p, item);
    }
    if (item >= p.n_items) return; // (the whole workgroup)
    // timeline: consumer wave 0 -> words 0..31, producer wave 8 -> 32..63 of the workgroup's slot ([0]/[31] entry / exit clock, [29]/[30] 100 MHz)
    unsigned long long *const dbg = (PS_TL(p.dbg) && blockIdx.x < 1024 && lane == 0 && (wave == 0 || wave == G4K_NC)) ? p.dbg + ((size_t)blockIdx.x * 2 + (wave == G4K_NC)) * 32 : nullptr;
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }
    if (wave >= G4K_NC) {
        g4k_producer_wave<EPI, G4K_NP, G4K_RING, WT>(p, item, lds, wave - G4K_NC, dbg);
        if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
        return;
    }
    const int ctl = wave & 3, uh = wave >> 2;
    int task, cb;
    g4k_item(p, item, task, cb);
    const int ct = cb * 4 + ctl;
    const int col = ct * 16 + m, colc = col < p.bs ? col : p.bs - 1; // (the last tile may be ragged: clamp the column metadata)
    const int ctc = ct * 16 < p.bs ? ct : (p.bs - 1) / 16; // (a wave past the batch walks the last tile's columns and stores nothing)
    const char *qf_ct = (const char *)p.qf + ((size_t)ctc * p.nsb << 13) + (size_t)(4 * uh) * 1024 + lane * 16;
    const uint8_t *mf_ct = p.mf + (size_t)ctc * p.nsb * 576;
    const int mc = colc & 15;
    const char *zero = lds + G4K_NST * G4K_STAGE;
    float *xch = (float *)(lds + G4K_XCH) + ((size_t)ctl * 64 + lane) * 48;
    int dbg_n = 1;
    auto mark = [&](int g) { if (dbg && (g < 8 || (g & 7) == 7) && dbg_n < 29) dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); };
    ps_u32x4 B[4];
#pragma unroll
    for (int k = 0; k < 4; k++) B[k] = *(const ps_u32x4 *)(qf_ct + k * 1024);
    const int boff = WT == PS_Q5_K ? (kb >> 1) * 16 : uh * 16;
    G4KMeta M = g4k_meta(mf_ct, 0, mc, boff);
    while (item < p.n_items) {
        g4k_item(p, item, task, cb);
        int wi, pair;
        (void)g4k_rows<EPI>(p, task, wi, pair);
        const G4KMat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
        G4KAcc T;
        T.clear();
        auto step = [&](const int sb) {
            const int nb = sb + 1 == p.nsb ? 0 : sb + 1; // (the next item meets the same columns from super-block 0)
            const G4KMeta Mn = g4k_meta(mf_ct, nb, mc, boff); // a step ahead, like B
            if (!(sb & 1)) __syncthreads(); // the producers have parked this step and the next
            mark(sb);
            g4k_superblock<WT>(T, lds + (sb & (G4K_NST - 1)) * G4K_STAGE, zero, B, qf_ct + ((size_t)nb << 13), M, m, kb, uh);
#ifdef G4K_MARK2
            if (dbg) { asm volatile("" : "+v"(T.acc[0][0][0]), "+v"(T.acc[1][3][1]), "+v"(T.accm[1][1][1])); mark(sb); } // (the step's chains done)
#endif
            M = Mn;
        };
        for (int sb = 0; sb < p.nsb - 1; sb++) step(sb);
        // what the epilogue adds is fetched a step early (clamped addresses, never a branch around a load).  The epilogue is
        // shared: accumulator half uh finishes rows 4 kb + 2 uh, + 1 of both tiles.
        float2 rsd[2], bia[2];
        if (EPI != 1) {
            const int cole = col < p.bs ? col : 0;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int64_t row0 = (int64_t)(2 * pair + t) * 16 + kb * 4 + 2 * uh;
                rsd[t] = *(const float2 *)((p.residual && wi == 0 ? p.residual : W.out) + (int64_t)cole * W.ldo + row0);
                bia[t] = *(const float2 *)((W.bias ? W.bias : W.out) + row0);
            }
        }
        step(p.nsb - 1);
        // ---- the two accumulator halves of a tile meet (hsum_float_8, ggml-quants.c:62-68, adds lane u to lane u + 4 first:
        // exactly the two halves; the acc_m reduction as row_reduce<PS_Q4_K> does): each hands the other the chains of the
        // rows the other finishes
        {
            float *mine = xch + uh * 24;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    // the other half's rows: 2 (1 - uh) + rr (selects, not a runtime index: the chains stay in registers)
#define G4K_SEL(fn, k) (uh ? T.fn(t, rr, k) : T.fn(t, 2 + rr, k))
                    *(float4 *)(mine + (t * 2 + rr) * 4) = make_float4(G4K_SEL(a, 0), G4K_SEL(a, 1), G4K_SEL(a, 2), G4K_SEL(a, 3));
                    *(float2 *)(mine + 16 + (t * 2 + rr) * 2) = make_float2(G4K_SEL(am, 0), G4K_SEL(am, 1));
#undef G4K_SEL
                }
        }
        __syncthreads(); // X: (a whole item of barriers lies between this exchange and the next one's stores)
        mark(0);
        {
            const float *theirs = xch + (1 - uh) * 24;
            float y[2][2];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
#define G4K_OWN(fn, k) (uh ? T.fn(t, 2 + rr, k) : T.fn(t, rr, k))
                    const float4 o4 = *(const float4 *)(theirs + (t * 2 + rr) * 4);
                    const float2 o2 = *(const float2 *)(theirs + 16 + (t * 2 + rr) * 2);
                    // lanes u < 4 (and mins lanes 0, 1) are half 0's, lanes u + 4 (mins 2, 3) half 1's
                    const float s0 = __fadd_rn(G4K_OWN(a, 0), o4.x), s1 = __fadd_rn(G4K_OWN(a, 1), o4.y), s2 = __fadd_rn(G4K_OWN(a, 2), o4.z), s3 = __fadd_rn(G4K_OWN(a, 3), o4.w);
                    const float res = __fadd_rn(__fadd_rn(s0, s2), __fadd_rn(s1, s3));
                    const float w0 = G4K_OWN(am, 0), w1 = G4K_OWN(am, 1);
                    const float ma = uh ? o2.x : w0, mb = uh ? o2.y : w1;   // acc_m lanes 0, 1
                    const float mc2 = uh ? w0 : o2.x, md = uh ? w1 : o2.y;  // acc_m lanes 2, 3
#undef G4K_OWN
                    const float mm = WT == PS_Q5_K ? ma : __fadd_rn(__fadd_rn(ma, mc2), __fadd_rn(mb, md)); // (Q5_K: hsum_float_8(acc) + summs)
                    y[t][rr] = __fadd_rn(res, mm);
                }
            if (col < p.bs) {
                if (EPI == 1) {
                    const int64_t row0 = (int64_t)task * 16 + kb * 4 + 2 * uh;
                    const uint64_t *tab = (const uint64_t *)(lds + G4K_TAB);
                    float o[2];
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) // ps_silu_mul with the table in LDS
                        o[rr] = __fmul_rn(__fmul_rn(y[0][rr], __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-y[0][rr], tab)))), y[1][rr]);
                    *(float2 *)(W.out + (int64_t)col * W.ldo + row0) = make_float2(o[0], o[1]);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int64_t row0 = (int64_t)(2 * pair + t) * 16 + kb * 4 + 2 * uh;
                        float v[2];
                        const float bv[2] = {bia[t].x, bia[t].y}, rv[2] = {rsd[t].x, rsd[t].y};
#pragma unroll
                        for (int rr = 0; rr < 2; rr++) {
                            v[rr] = y[t][rr];
                            if (W.bias) v[rr] = __fadd_rn(v[rr], bv[rr]);
                            if (p.residual && wi == 0) v[rr] = __fadd_rn(rv[rr], v[rr]);
                        }
                        g4k_store_pair<0>(p, wi, W, col, row0, v[0], v[1]); // (the RoPE epilogue lives in the narrow kernel only: here it costs 9 spills at the 168-register cap and gives the saved launch back, 14.74 vs 14.81 k tok/s)
                    }
                }
            }
        }
        mark(0);
        item = g4k_next_item(p, item);
    }
    if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- narrow batches (at most 32 columns: tree verify, prompt tails).  With one or two live column tiles the wide mapping
// leaves most consumers walking dead columns, and a consumer's step is a chain of LDS and L2 round trips (0.83 us) that does
// not get shorter when others idle.  Here the eight consumers split ONE tile's work eight ways (CT = 1: wave = accumulator
// lane u, waves 0..3 also mins lane v = u) or two tiles' four ways (CT = 2: two lanes u per wave, mins lane v = group): two
// or four matrix instructions and a handful of chains per wave and step, and the B / column-metadata ring runs FOUR steps
// ahead (a short step would otherwise wait for L2), so the workgroup runs at the producers' pace.  Same producers, same LDS
// stages, same persistent items; every wave parks its chains at the end of an item and the waves of groups 0 and 1 finish
// rows 4 kb + 0, 1 and 4 kb + 2, 3 (hsum_float_8's order) from LDS.
constexpr int G4K_RINGN = 4; // (a ring of eight super-blocks measured slower: 12 wide 5.09 vs 4.78 ms)
constexpr int G4K_NPN = 8; // producing waves of the narrow kernel: one accumulator lane each (the consumers are light there, the producers set the pace)
template <int EPI, int CT, int WT>
__global__ __launch_bounds__((G4K_NC + G4K_NPN) * 64) void gemm4k_narrow_kernel(const G4KParams p) {
    constexpr int NU = CT, XW = CT == 1 ? 2 : 4; // accumulator lanes per wave; floats per (tile, row) in the exchange (NU chains + 1 mins chain, padded)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (threadIdx.x < 8) ((uint32_t *)(lds + G4K_NST * G4K_STAGE))[threadIdx.x] = 0u; // the zero operands (visible after the first barrier)
    constexpr int TAB = G4K_XCH + 8 * 64 * 8 * XW * 4;
    if (EPI == 1 && threadIdx.x >= 64 && threadIdx.x < 64 + PS_EXP2F_N) ((uint64_t *)(lds + TAB))[threadIdx.x - 64] = ps_exp2f_tab[threadIdx.x - 64];
    int item = (int)blockIdx.x;
    {
        int t, c;
        g4k_item(p, item, t, c);
        if (t >= p.n_tasks) item = g4k_next_item(p, item);
    }
    if (item >= p.n_items) return; // (the whole workgroup)
    unsigned long long *const dbg = (PS_TL(p.dbg) && blockIdx.x < 1024 && lane == 0 && (wave == 0 || wave == G4K_NC)) ? p.dbg + ((size_t)blockIdx.x * 2 + (wave == G4K_NC)) * 32 : nullptr;
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }
    if (wave >= G4K_NC) {
        g4k_producer_wave<EPI, G4K_NPN, G4K_RINGN, WT>(p, item, lds, wave - G4K_NC, dbg);
        if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
        return;
    }
    const int ctl = wave % CT, ug = wave / CT; // (column tile, accumulator-lane group: lanes u = NU ug .. NU ug + NU - 1)
    const bool has_v = WT == PS_Q5_K ? ug == 0 : ug < 4; // mins lane v = ug (Q5_K: group 0 keeps the one scalar chain, k-group kb supplying lane v = kb)
    int task, cb;
    g4k_item(p, item, task, cb);
    const int ct = cb * CT + ctl;
    const int col = ct * 16 + m, colc = col < p.bs ? col : p.bs - 1;
    const int ctc = ct * 16 < p.bs ? ct : (p.bs - 1) / 16;
    const char *qf_ct = (const char *)p.qf + ((size_t)ctc * p.nsb << 13) + (size_t)(NU * ug) * 1024 + lane * 16;
    const uint8_t *mf_ct = p.mf + (size_t)ctc * p.nsb * 576 + (colc & 15) * 4;                          // the column's scale
    const uint8_t *ms_ct = p.mf + (size_t)ctc * p.nsb * 576 + 64 + (colc & 15) * 32 + (WT == PS_Q5_K ? kb : (has_v ? ug : 0)) * 8; // the four 16-sums of mins lane v
    const char *zero = lds + G4K_NST * G4K_STAGE;
    float *xall = (float *)(lds + G4K_XCH);
    const g4k_f4 zf = {0.f, 0.f, 0.f, 0.f};
    // the ring: operands of the next four super-blocks
    ps_u32x4 B[4][NU];
    float yd[4];
    uint2 b16[4];
    auto ring_load = [&](const int j, const int sb) {
#pragma unroll
        for (int k = 0; k < NU; k++) B[j][k] = *(const ps_u32x4 *)(qf_ct + ((size_t)sb << 13) + k * 1024);
        yd[j] = *(const float *)(mf_ct + (size_t)sb * 576);
        b16[j] = *(const uint2 *)(ms_ct + (size_t)sb * 576);
    };
#pragma unroll
    for (int j = 0; j < 4; j++) ring_load(j, j); // (nsb >= 4)
    int nx = 4 == p.nsb ? 0 : 4; // the super-block the ring fetches next (wraps: the next item meets the same columns from super-block 0)
    while (item < p.n_items) {
        g4k_item(p, item, task, cb);
        int wi, pair;
        (void)g4k_rows<EPI>(p, task, wi, pair);
        const G4KMat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
        float acc[2][4][NU], accm[2][4];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int k = 0; k < NU; k++) acc[t][r][k] = 0.f;
                accm[t][r] = 0.f;
            }
#pragma clang loop unroll(disable)
        for (int sb0 = 0; sb0 < p.nsb; sb0 += 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!(j & 1)) __syncthreads(); // the producers have parked this step and the next
                int so = j * G4K_STAGE; // (sb0 % 4 == 0: stage sb % 4 = j)
                asm volatile("" : "+s"(so)); // (opaque: or the addresses of all four stages are hoisted out of the loop and spilled)
                const char *st = lds + so;
                float dr[2][4], dmin[2][4];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const g4k_f4 dda = *(const g4k_f4 *)(st + G4K_DD + (16 * t + 4 * kb) * 8), ddb = *(const g4k_f4 *)(st + G4K_DD + (16 * t + 4 * kb) * 8 + 16);
                    dr[t][0] = __fmul_rn(yd[j], dda[0]); dr[t][1] = __fmul_rn(yd[j], dda[2]); dr[t][2] = __fmul_rn(yd[j], ddb[0]); dr[t][3] = __fmul_rn(yd[j], ddb[2]);
                    dmin[t][0] = __fmul_rn(-yd[j], dda[1]); dmin[t][1] = __fmul_rn(-yd[j], dda[3]); dmin[t][2] = __fmul_rn(-yd[j], ddb[1]); dmin[t][3] = __fmul_rn(-yd[j], ddb[3]);
                }
                const char *ap = st + g4k_plane(kb) + m * G4K_RS + (NU * ug) * 16;
#pragma unroll
                for (int k = 0; k < NU; k++) {
                    g4k_h8 bv;
                    __builtin_memcpy(&bv, &B[j][k], 16);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const ps_u32x4 ao = *(const ps_u32x4 *)(ap + t * 16 * G4K_RS + k * 16);
                        g4k_h8 av;
                        __builtin_memcpy(&av, &ao, 16);
                        const g4k_f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, zf, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[t][r][k] = __fmaf_rn(dr[t][r], si[r], acc[t][r][k]);
                    }
                }
                { // (waves without a mins lane read the zero operands: their chains stay 0 and nobody reads them)
                    g4k_h2 g0, g1;
                    __builtin_memcpy(&g0, &b16[j].x, 4); __builtin_memcpy(&g1, &b16[j].y, 4);
                    const g4k_h4 bm = {g0[0], g0[1], g1[0], g1[1]};
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const uint2 ma = *(const uint2 *)(WT == PS_Q5_K ? (has_v ? st + G4K_MINS + (16 * t + m) * 32 + kb * 8 : zero)
                                                                          : (kb == 0 && has_v ? st + G4K_MINS + (16 * t + m) * 32 + ug * 8 : zero));
                        g4k_h2 a0, a1;
                        __builtin_memcpy(&a0, &ma.x, 4); __builtin_memcpy(&a1, &ma.y, 4);
                        const g4k_h4 am = {a0[0], a0[1], a1[0], a1[1]};
                        const g4k_f4 pr = __builtin_amdgcn_mfma_f32_16x16x16f16(am, bm, zf, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; r++) accm[t][r] = WT == PS_Q5_K ? __fadd_rn(accm[t][r], __fmul_rn(dmin[t][r], pr[r])) : __fmaf_rn(dmin[t][r], pr[r], accm[t][r]);
                    }
                }
                ring_load(j, nx);
                nx = nx + 1 == p.nsb ? 0 : nx + 1;
            }
        }
        // ---- every wave parks its chains; groups 0 and 1 finish rows 4 kb + 2 ug, + 1 of both tiles
        {
            float *mine = xall + ((size_t)wave * 64 + lane) * (8 * XW);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (NU == 1) *(float2 *)(mine + (t * 4 + r) * XW) = make_float2(acc[t][r][0], accm[t][r]);
                    else *(float4 *)(mine + (t * 4 + r) * XW) = make_float4(acc[t][r][0], acc[t][r][NU - 1], accm[t][r], 0.f);
                }
        }
        __syncthreads(); // X
        if (ug < 2) {
            float y[2][2];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int r = 2 * ug + rr;
                    float au[8], mv[4];
#pragma unroll
                    for (int u = 0; u < 8; u++) // lane u lives in wave (ctl, group u / NU), slot u % NU
                        au[u] = xall[((size_t)(ctl + CT * (u / NU)) * 64 + lane) * (8 * XW) + (t * 4 + r) * XW + (u % NU)];
#pragma unroll
                    for (int v = 0; v < 4; v++) // mins lane v lives in group v, after the NU chains
                        mv[v] = xall[((size_t)(ctl + CT * v) * 64 + lane) * (8 * XW) + (t * 4 + r) * XW + NU];
                    const float s0 = __fadd_rn(au[0], au[4]), s1 = __fadd_rn(au[1], au[5]), s2 = __fadd_rn(au[2], au[6]), s3 = __fadd_rn(au[3], au[7]);
                    const float res = __fadd_rn(__fadd_rn(s0, s2), __fadd_rn(s1, s3));
                    const float mm = WT == PS_Q5_K ? mv[0] : __fadd_rn(__fadd_rn(mv[0], mv[2]), __fadd_rn(mv[1], mv[3]));
                    y[t][rr] = __fadd_rn(res, mm);
                }
            if (col < p.bs) {
                if (EPI == 1) {
                    const int64_t row0 = (int64_t)task * 16 + kb * 4 + 2 * ug;
                    const uint64_t *tab = (const uint64_t *)(lds + TAB);
                    float o[2];
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) // ps_silu_mul with the table in LDS
                        o[rr] = __fmul_rn(__fmul_rn(y[0][rr], __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-y[0][rr], tab)))), y[1][rr]);
                    *(float2 *)(W.out + (int64_t)col * W.ldo + row0) = make_float2(o[0], o[1]);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int64_t row0 = (int64_t)(2 * pair + t) * 16 + kb * 4 + 2 * ug;
                        float v[2];
#pragma unroll
                        for (int rr = 0; rr < 2; rr++) {
                            v[rr] = y[t][rr];
                            if (W.bias) v[rr] = __fadd_rn(v[rr], W.bias[row0 + rr]);
                            if (p.residual && wi == 0) v[rr] = __fadd_rn(p.residual[(int64_t)col * W.ldo + row0 + rr], v[rr]);
                        }
                        g4k_store_pair<2>(p, wi, W, col, row0, v[0], v[1]);
                    }
                }
            }
        }
        item = g4k_next_item(p, item);
    }
    if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}

// ================================================================ Q6_K (the Q4_K_M / Q5_K_M mixes: attn_v, half of the ffn_down, output)
// ggml_vec_dot_q6_K_q8_K (AVX2, libs/ggml/src/ggml-quants.c:8713-8790) has the same shape: per super-block one int32 per
// accumulator lane u,  sumi[u] = sum over the eight 32-element vectors g of  scale[g][u / 4] * dot4((q6 - 32)[g][4u..], y[g][4u..])
// (the -32 is applied through the q8 sums there: the same integer), then  acc[u] = fma(d * y.d, (float)sumi[u], acc[u])  and
// hsum_float_8 -- no mins term.  The B operand is the SAME fragment-major fp16 copy of the Q8_K quants.  |(q6 - 32) * scale|
// reaches 32 * 128 = 4096, past the 2048 up to which fp16 holds every integer, so the int8 scale is split  scale = even + odd
// (even = scale & ~1, odd = scale & 1):  A_hi = (q6 - 32) * even  is an EVEN integer <= 4096 (exact in fp16),  A_lo = (q6 - 32) or 0;
// two MFMAs on the same B, the second accumulating onto the first (every partial sum < 32 * 4096 * 127 < 2^24: exact).
// Workgroup = the wide Q4_K form (two row tiles x 64 columns, 4 column tiles x 2 accumulator halves, four producers,
// persistent items); two LDS stages of (A_hi, A_lo planes, d), one barrier per step.
struct G6KParams {
    const uint8_t *ql, *qh;   // [N][nsb][u][16 B], [N][nsb][u][8 B]   (ps_internal.h)
    const int8_t *sc;         // [N][nsb][16]
    const uint16_t *d;        // [N][nsb] fp16
    float *out;
    const float *bias, *residual;
    int64_t N, ldo;
    int nsb, bs, n_tasks, n_cb, n_items;
    const _Float16 *qf;
    const uint8_t *mf;
};
constexpr int G6K_PLANE = 4 * G4K_KB + G4K_PAD, G6K_DD = 2 * G6K_PLANE, G6K_STAGE = G6K_DD + 32 * 4, G6K_NST = 2;
constexpr int G6K_XCH = G6K_NST * G6K_STAGE, G6K_LDS = G6K_XCH + 2 * 4 * 64 * 16 * 4; // exchange: [half][column tile][lane][16 floats]

__device__ __forceinline__ void g6k_item(const G6KParams &p, const int i, int &task, int &cb) {
    cb = (i >> 3) % p.n_cb;
    task = (i / (8 * p.n_cb)) * 8 + (i & 7);
}
__device__ __forceinline__ int g6k_next_item(const G6KParams &p, int i) {
    for (i += (int)gridDim.x; i < p.n_items; i += (int)gridDim.x) {
        int t, c;
        g6k_item(p, i, t, c);
        if (t < p.n_tasks) break;
    }
    return i < p.n_items ? i : p.n_items;
}

// producer wave hw: accumulator lanes u = 2 hw, 2 hw + 1 of the 32 rows; lane = (row l / 2, half p = l % 2 of the super-block:
// k-groups kb = 2p, 2p + 1 = vectors 4p .. 4p + 3)
__device__ __forceinline__ void g6k_produce(const uint2 (&q)[2], const uint32_t (&hq)[2], const uint2 scl, const uint32_t dh, char *st, const int row,
                                            const int hw, const int p) {
    const g4k_h2 k1056 = {(_Float16)1056.f, (_Float16)1056.f};
    const int ush = 8 * (hw >> 1); // the 16-block of u within a vector: u / 4 = hw / 2
#pragma unroll
    for (int e = 0; e < 2; e++) { // kb = 2p + e: vectors g0 = 2 kb (scale bytes 0 / 1 of the pair), g1 = 2 kb + 1 (bytes 2 / 3)
        const uint32_t sw = e ? scl.y : scl.x;
        const int s0i = (int)(int8_t)(sw >> ush), s1i = (int)(int8_t)(sw >> (16 + ush));
        const _Float16 e0 = (_Float16)(float)(s0i & ~1), e1 = (_Float16)(float)(s1i & ~1);
        const g4k_h2 ev0 = {e0, e0}, ev1 = {e1, e1};
        const uint32_t om0 = (s0i & 1) ? 0xffffffffu : 0u, om1 = (s1i & 1) ? 0xffffffffu : 0u;
#pragma unroll
        for (int j = 0; j < 2; j++) { // u = 2 hw + j
            const uint32_t w0 = q[j].x, w1 = q[j].y, h = hq[j];
            // 6-bit values as fp16 (1024 + v): (e0, e2), (e1, e3) of vector g0, then of g1; minus 1056 = v - 32, exact
            const uint32_t t[4] = {((w0 >> (4 * e)) & 0x000F000Fu) | (((h >> (4 * e)) & 0x00030003u) << 4) | 0x64006400u,
                                   ((w0 >> (4 * e + 8)) & 0x000F000Fu) | (((h >> (4 * e + 8)) & 0x00030003u) << 4) | 0x64006400u,
                                   ((w1 >> (4 * e)) & 0x000F000Fu) | (((h >> (4 * e + 2)) & 0x00030003u) << 4) | 0x64006400u,
                                   ((w1 >> (4 * e + 8)) & 0x000F000Fu) | (((h >> (4 * e + 10)) & 0x00030003u) << 4) | 0x64006400u};
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                g4k_h2 v;
                __builtin_memcpy(&v, &t[k], 4);
                v = v - k1056;
                uint32_t vb;
                __builtin_memcpy(&vb, &v, 4);
                lo[k] = vb & (k < 2 ? om0 : om1);
                v = v * (k < 2 ? ev0 : ev1); // (even, |.| <= 4096: exact)
                __builtin_memcpy(&hi[k], &v, 4);
            }
            char *dst = st + g4k_plane(2 * p + e) + row * G4K_RS + (2 * hw + j) * 16;
            *(uint4 *)dst = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *(uint4 *)(dst + G6K_PLANE) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
    if (hw == 1 && p == 0) *(float *)(st + G6K_DD + row * 4) = ps_h2f((uint16_t)dh);
}

__device__ __forceinline__ void g6k_producer_wave(const G6KParams &p, int item, char *lds, const int hw) {
    const int lane = threadIdx.x & 63, nsb = p.nsb;
    const int row = lane >> 1, pp = lane & 1;
    const uint8_t *qb, *hb, *sb_, *db;
    auto point = [&](int it) {
        int task, cb;
        g6k_item(p, it, task, cb);
        const int64_t gr = (int64_t)task * 32 + row; // (two consecutive row tiles)
        qb = p.ql + (gr * nsb * 8 + 2 * hw) * 16 + pp * 8;
        hb = p.qh + (gr * nsb * 8 + 2 * hw) * 8 + pp * 4;
        sb_ = (const uint8_t *)p.sc + gr * nsb * 16 + pp * 8;
        db = (const uint8_t *)p.d + gr * nsb * 2;
    };
    uint2 rq[G4K_RING][2], rs[G4K_RING];
    uint32_t rh[G4K_RING][2], rd[G4K_RING];
    auto load = [&](int g, uint2 (&a)[2], uint32_t (&h)[2], uint2 &sc, uint32_t &dh) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            a[j] = *(const uint2 *)(qb + (size_t)g * 128 + j * 16);
            h[j] = *(const uint32_t *)(hb + (size_t)g * 64 + j * 8);
        }
        sc = *(const uint2 *)(sb_ + (size_t)g * 16);
        dh = *(const uint16_t *)(db + (size_t)g * 2);
    };
    point(item);
#pragma unroll
    for (int k = 0; k < G4K_RING; k++) load(k, rq[k], rh[k], rs[k], rd[k]);
    int c_item = item, c_g = G4K_RING;
    bool first = true;
    while (item < p.n_items) {
        for (int g0 = 0; g0 < nsb; g0 += G4K_RING) {
            if (c_g == nsb) {
                const int nx = g6k_next_item(p, c_item);
                if (nx < p.n_items) { c_item = nx; c_g = 0; point(nx); } else c_g = nsb - G4K_RING;
            }
#pragma unroll
            for (int k = 0; k < G4K_RING; k++) {
                uint2 a[2] = {rq[k][0], rq[k][1]};
                uint32_t h[2] = {rh[k][0], rh[k][1]};
                const uint2 sc = rs[k];
                const uint32_t dh = rd[k];
                load(c_g + k, rq[k], rh[k], rs[k], rd[k]);
                g6k_produce(a, h, sc, dh, lds + ((g0 + k) & 1) * G6K_STAGE, row, hw, pp);
                if (g0 == 0 && k == 0 && !first) __syncthreads(); // X of the previous item
                __syncthreads(); // stage g0 + k is parked
            }
            c_g += G4K_RING;
        }
        first = false;
        item = g6k_next_item(p, item);
    }
    __syncthreads(); // X of the last item
}

__global__ __launch_bounds__((G4K_NC + G4K_NP) * 64) void gemm6k_kernel(const G6KParams p) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int item = (int)blockIdx.x;
    {
        int t, c;
        g6k_item(p, item, t, c);
        if (t >= p.n_tasks) item = g6k_next_item(p, item);
    }
    if (item >= p.n_items) return;
    if (wave >= G4K_NC) { g6k_producer_wave(p, item, lds, wave - G4K_NC); return; }
    const int ctl = wave & 3, uh = wave >> 2;
    int task, cb;
    g6k_item(p, item, task, cb);
    const int ct = cb * 4 + ctl;
    const int col = ct * 16 + m, colc = col < p.bs ? col : p.bs - 1;
    const int ctc = ct * 16 < p.bs ? ct : (p.bs - 1) / 16;
    const char *qf_ct = (const char *)p.qf + ((size_t)ctc * p.nsb << 13) + (size_t)(4 * uh) * 1024 + lane * 16;
    const uint8_t *mf_ct = p.mf + (size_t)ctc * p.nsb * 576 + (colc & 15) * 4;
    float *xch = (float *)(lds + G6K_XCH) + ((size_t)ctl * 64 + lane) * 16;
    const g4k_f4 zf = {0.f, 0.f, 0.f, 0.f};
    ps_u32x4 B[4];
#pragma unroll
    for (int k = 0; k < 4; k++) B[k] = *(const ps_u32x4 *)(qf_ct + k * 1024);
    float yd = *(const float *)mf_ct;
    while (item < p.n_items) {
        g6k_item(p, item, task, cb);
        float acc[2][4][4];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[t][r][k] = 0.f;
        for (int sb = 0; sb < p.nsb; sb++) {
            const int nb = sb + 1 == p.nsb ? 0 : sb + 1;
            const float ydn = *(const float *)(mf_ct + (size_t)nb * 576);
            __syncthreads(); // the producers have parked this step
            const char *st = lds + (sb & 1) * G6K_STAGE;
            float dr[2][4];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const g4k_f4 dd = *(const g4k_f4 *)(st + G6K_DD + (16 * t + 4 * kb) * 4); // d of rows 4 kb + r
#pragma unroll
                for (int r = 0; r < 4; r++) dr[t][r] = __fmul_rn(yd, dd[r]);
            }
            const char *ap = st + g4k_plane(kb) + m * G4K_RS + (4 * uh) * 16;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                g4k_h8 bv;
                __builtin_memcpy(&bv, &B[k], 16);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const ps_u32x4 ah = *(const ps_u32x4 *)(ap + t * 16 * G4K_RS + k * 16), al = *(const ps_u32x4 *)(ap + G6K_PLANE + t * 16 * G4K_RS + k * 16);
                    g4k_h8 avh, avl;
                    __builtin_memcpy(&avh, &ah, 16); __builtin_memcpy(&avl, &al, 16);
                    g4k_f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(avh, bv, zf, 0, 0, 0);
                    si = __builtin_amdgcn_mfma_f32_16x16x32_f16(avl, bv, si, 0, 0, 0); // (float)sumi[4 uh + k] of rows 4 kb + r, this lane's column
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[t][r][k] = __fmaf_rn(dr[t][r], si[r], acc[t][r][k]);
                }
                B[k] = *(const ps_u32x4 *)(qf_ct + ((size_t)nb << 13) + k * 1024);
            }
            yd = ydn;
        }
        // ---- the halves meet (hsum_float_8: lane u + lane u + 4 first); half uh finishes rows 4 kb + 2 uh, + 1 of both tiles
        {
            float *mine = xch + uh * 16 * 64 * 4; // (second half of the exchange area)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
#define G6K_SEL(k) (uh ? acc[t][rr][k] : acc[t][2 + rr][k])
                    *(float4 *)(mine + (t * 2 + rr) * 4) = make_float4(G6K_SEL(0), G6K_SEL(1), G6K_SEL(2), G6K_SEL(3));
#undef G6K_SEL
                }
        }
        __syncthreads(); // X
        {
            const float *theirs = xch + (1 - uh) * 16 * 64 * 4;
            if (col < p.bs) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int64_t row0 = (int64_t)task * 32 + t * 16 + kb * 4 + 2 * uh;
                    float v[2];
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        const float4 o4 = *(const float4 *)(theirs + (t * 2 + rr) * 4);
#define G6K_OWN(k) (uh ? acc[t][2 + rr][k] : acc[t][rr][k])
                        const float s0 = __fadd_rn(G6K_OWN(0), o4.x), s1 = __fadd_rn(G6K_OWN(1), o4.y), s2 = __fadd_rn(G6K_OWN(2), o4.z), s3 = __fadd_rn(G6K_OWN(3), o4.w);
#undef G6K_OWN
                        v[rr] = __fadd_rn(__fadd_rn(s0, s2), __fadd_rn(s1, s3));
                        if (p.bias) v[rr] = __fadd_rn(v[rr], p.bias[row0 + rr]);
                        if (p.residual) v[rr] = __fadd_rn(p.residual[(int64_t)col * p.ldo + row0 + rr], v[rr]);
                    }
                    *(float2 *)(p.out + (int64_t)col * p.ldo + row0) = make_float2(v[0], v[1]);
                }
            }
        }
        item = g6k_next_item(p, item);
    }
}

} // namespace

int g_g4k_cbx = getenv("PS_G4K_CBX") ? atoi(getenv("PS_G4K_CBX")) : 4; // ps_hip_debug_set(6, v): column blocks per XCD of the wide Q4_K / Q5_K mat-mul's item order (0: round 2's; default 4: profiles/r05_g4k_item_order.txt)
int g_g4k_par = getenv("PS_GEMM4K_PAR") ? atoi(getenv("PS_GEMM4K_PAR")) : 1; // ps_hip_debug_set(3, v): the few-tile narrow-batch form (gemm4k_par_kernel)
// grid, persistence and the wide / narrow choice for a filled-in G4KParams (tasks, pointers, wt)
// ---- narrow batches, wave-autonomous form (round 3): at most 16 columns, Q4_K.  The producer / consumer kernels above pay a
// fixed ~0.45 us per super-block step (barriers, LDS round trips) whatever the width, and a K walk is sequential: 12 columns cost
// 39.6 us for the gate / up launch (1.7 TB/s) and 31 us for down, where 128 tasks leave half the chip idle
// (profiles/r03_tree12_kernel_stats.txt).  Here ONE WAVE owns one 16-row tile -- all eight accumulator lanes and all four mins
// lanes -- for the whole K walk: no stage shared with other waves, no barrier.  Lane (row m, k-group kb) needs dword kb of the
// 16 bytes the lane-major layout keeps for (row, u); fetched as dwords that is eight loads per super-block over the same sixteen
// 128-byte lines, every one of them served by L2 (measured: 17 TB/s of L2 reads, slower than the staged kernel).  So a wave fetches
// each line ONCE -- lane (r, u) its 16 bytes, two fully coalesced 1 KiB loads per super-block -- and turns the 2 KiB around in
// a private LDS buffer (padded rows: conflict-free both ways; one wave's LDS operations are ordered).  The wave builds its fp16 A operands in registers (g4k_produce's arithmetic),
// fetches the B fragments from L2 as it goes, runs its 48 fp32 chains and finishes hsum_float_8 in its own registers; a
// register ring RA super-blocks deep covers the weight latency.  EPI 1: two waves per workgroup (gate tile, up tile), one
// exchange through LDS at the end.
template <int EPI, int RA, int MB> // MB: waves per SIMD the register allocation leaves room for
__global__ __launch_bounds__(EPI == 1 ? 128 : 64, MB) void gemm4k_wav_kernel(const G4KParams p) {
    const int wave = EPI == 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0, lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    __shared__ float xup[EPI == 1 ? 64 * 4 : 1];
    __shared__ uint64_t tab[EPI == 1 ? PS_EXP2F_N : 1];
    if (EPI == 1 && threadIdx.x < PS_EXP2F_N) tab[threadIdx.x] = ps_exp2f_tab[threadIdx.x];
    const int nsb = p.nsb;
    int wi, pair;
    const int task = EPI == 1 ? (int)blockIdx.x : (int)(blockIdx.x >> 1);
    const G4KRows R = g4k_rows<EPI>(p, task, wi, pair);
    const G4KMat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
    const int tt = EPI == 1 ? wave : (int)(blockIdx.x & 1);
    const int tile = tt ? R.tile[1] : R.tile[0];
    const size_t g8 = (size_t)2 * tile + (m >> 3); // the 8-row group of this lane's row
    const uint8_t *qb = (tt ? R.qs[1] : R.qs[0]) + ((size_t)2 * tile * nsb << 10) + (size_t)lane * 16; // the tile's two 1 KiB units of a super-block, 16 B per lane
    const uint8_t *hb = (tt ? R.aux[1] : R.aux[0]) + g8 * nsb * 128 + (size_t)(m & 7) * 16;
    // the wave's transposition buffer: row (16) x (8 u x 4 dwords), rows padded to 36 dwords (conflict-free both ways)
    __shared__ uint32_t trb[EPI == 1 ? 2 : 1][16 * 36];
    uint32_t *const trw = trb[wave];
    const int col = m, colc = col < p.bs ? col : p.bs - 1;
    const char *qfp = (const char *)p.qf + lane * 16; // (one column tile: ct = 0)
    const uint8_t *ydp = p.mf + colc * 4, *b16p = p.mf + 64 + colc * 32;

    ps_u32x4 rq[RA][2], rh[RA], rb[8], rs[2][2]; // weights and headers RA steps ahead; fragments one step ahead; the column's 16-sums two
    float ryd[2];
    auto load_a = [&](const int s, const int sb) { // every line of the tile ONCE: lane (r, u) takes its 16 bytes of both 8-row groups
        rq[s][0] = __builtin_nontemporal_load((const ps_u32x4 *)(qb + ((size_t)sb << 10)));
        rq[s][1] = __builtin_nontemporal_load((const ps_u32x4 *)(qb + ((size_t)(nsb + sb) << 10)));
        rh[s] = *(const ps_u32x4 *)(hb + (size_t)sb * 128);
    };
    auto load_m = [&](const int s, const int sb) {
        ryd[s]   = *(const float *)(ydp + (size_t)sb * 576);
        rs[s][0] = *(const ps_u32x4 *)(b16p + (size_t)sb * 576);
        rs[s][1] = *(const ps_u32x4 *)(b16p + (size_t)sb * 576 + 16);
    };
#pragma unroll
    for (int s = 0; s < RA; s++) load_a(s, s < nsb ? s : nsb - 1);
    load_m(0, 0); load_m(1, 1 < nsb ? 1 : 0);
#pragma unroll
    for (int u = 0; u < 8; u++) rb[u] = *(const ps_u32x4 *)(qfp + u * 1024);

    float acc[4][8], accm[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) acc[r][u] = 0.f;
#pragma unroll
        for (int v = 0; v < 4; v++) accm[r][v] = 0.f;
    }
    const g4k_f4 zf = {0.f, 0.f, 0.f, 0.f};
    const g4k_h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f}, km1024 = {(_Float16)-1024.f, (_Float16)-1024.f};
    const uint32_t sel0 = 0x04000400u | ((uint32_t)(2 * (kb & 1)) * 0x00010001u), sel1 = sel0 + 0x00010001u; // scales 2 (kb & 1), + 1 of the four
    static_assert(RA % 2 == 0, "the 16-sum ring is two deep");
#pragma clang loop unroll(disable)
    for (int sb0 = 0; sb0 < nsb; sb0 += RA) { // (nsb % RA == 0: the host checks)
#pragma unroll
        for (int s = 0; s < RA; s++) {
            const int sb = sb0 + s, nx = sb + 1 < nsb ? sb + 1 : sb; // nx: the super-block whose fragments are requested now
            const float yd = ryd[s & 1];
            const ps_u32x4 h = rh[s];
            // the scales of sub-blocks 2 kb, 2 kb + 1 (get_scale_min_k4) as fp16 pairs (s, s) and (-1024 s, -1024 s)
            const uint32_t scb = (kb & 2) ? ((h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4)) : (h.y & 0x3f3f3f3fu);
            const uint32_t t0 = __builtin_amdgcn_perm(0x64646464u, scb, sel0), t1 = __builtin_amdgcn_perm(0x64646464u, scb, sel1);
            g4k_h2 s0, s1;
            __builtin_memcpy(&s0, &t0, 4); __builtin_memcpy(&s1, &t1, 4);
            s0 = s0 - k1024; s1 = s1 - k1024; // (bytes (s, 0x64) = fp16 1024 + s)
            const g4k_h2 n0 = s0 * km1024, n1 = s1 * km1024;
            // d, dmin of the rows this lane's results belong to (rows 4 kb + r: their headers sit in lanes 4 kb + r)
            float dr[4], dmn[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t hx = (uint32_t)__shfl((int)h.x, 4 * kb + r, 64);
                dr[r]  = __fmul_rn(yd, ps_h2f((uint16_t)(hx & 0xffff)));
                dmn[r] = __fmul_rn(-yd, ps_h2f((uint16_t)(hx >> 16)));
            }
            // lane (r, u) parks its 16 bytes (dwords kb = 0 .. 3); lane (m, kb) picks dword kb of (row m, u) for the eight u.  One
            // wave, LDS operations in order: no barrier
            *(ps_u32x4 *)(trw + (lane >> 3) * 36 + (lane & 7) * 4) = rq[s][0];
            *(ps_u32x4 *)(trw + (8 + (lane >> 3)) * 36 + (lane & 7) * 4) = rq[s][1];
            uint32_t wq[8];
#pragma unroll
            for (int u = 0; u < 8; u++) wq[u] = trw[m * 36 + u * 4 + kb];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t w = wq[u];
                // nibble pairs as fp16 (1024 + n): (e0, e2), (e1, e3) of sub-block 2 kb, then of 2 kb + 1;  fma(1024 + n, s, -1024 s) = n s, exact
                const uint32_t tq[4] = {(w & 0x000F000Fu) | 0x64006400u, ((w >> 8) & 0x000F000Fu) | 0x64006400u,
                                        ((w >> 4) & 0x000F000Fu) | 0x64006400u, ((w >> 12) & 0x000F000Fu) | 0x64006400u};
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    g4k_h2 v;
                    __builtin_memcpy(&v, &tq[k], 4);
                    v = __builtin_elementwise_fma(v, k < 2 ? s0 : s1, k < 2 ? n0 : n1);
                    __builtin_memcpy(&o[k], &v, 4);
                }
                const ps_u32x4 ao = {o[0], o[1], o[2], o[3]};
                g4k_h8 av, bv;
                __builtin_memcpy(&av, &ao, 16); __builtin_memcpy(&bv, &rb[u], 16);
                const g4k_f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, zf, 0, 0, 0); // (float)sumi[u] of rows 4 kb + r, this lane's column
                rb[u] = *(const ps_u32x4 *)(qfp + ((size_t)nx << 13) + u * 1024);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r][u] = __fmaf_rn(dr[r], si[r], acc[r][u]);
                if (u & 1) __builtin_amdgcn_sched_barrier(0); // two accumulator lanes at a time: interleaving all eight costs registers, hides nothing other waves do not
            }
            { // the four mins lanes: A = (m[2v], m[2v], m[2v+1], m[2v+1]) in the lanes of k-group 0, zero elsewhere; B = the column's 16-sums
                const uint32_t mn[2] = {h.z & 0x3f3f3f3fu, ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4)};
                const uint32_t bw[8] = {rs[s & 1][0].x, rs[s & 1][0].y, rs[s & 1][0].z, rs[s & 1][0].w, rs[s & 1][1].x, rs[s & 1][1].y, rs[s & 1][1].z, rs[s & 1][1].w};
                const g4k_h2 z2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, mn[v >> 1], 0x04000400u | (uint32_t)(((2 * v) & 3) * 0x00010001u));
                    const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, mn[v >> 1], 0x04000400u | (uint32_t)((((2 * v) & 3) + 1) * 0x00010001u));
                    g4k_h2 h0, h1;
                    __builtin_memcpy(&h0, &p0, 4); __builtin_memcpy(&h1, &p1, 4);
                    h0 = h0 - k1024; h1 = h1 - k1024;
                    if (kb != 0) { h0 = z2; h1 = z2; }
                    const g4k_h4 am = {h0[0], h0[1], h1[0], h1[1]};
                    g4k_h4 bm;
                    { const ps_u32x2 b2 = {bw[2 * v], bw[2 * v + 1]}; __builtin_memcpy(&bm, &b2, 8); }
                    const g4k_f4 pr = __builtin_amdgcn_mfma_f32_16x16x16f16(am, bm, zf, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) accm[r][v] = __fmaf_rn(dmn[r], pr[r], accm[r][v]);
                }
            }
            // the slots just used take the loads of RA (2) steps ahead; past the end they re-read the last super-block
            load_a(s, sb + RA < nsb ? sb + RA : nsb - 1);
            load_m(s & 1, sb + 2 < nsb ? sb + 2 : nsb - 1);
        }
    }
    // ---- hsum_float_8 (+ the mins chains) of rows 4 kb + r, in registers
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float s0 = __fadd_rn(acc[r][0], acc[r][4]), s1 = __fadd_rn(acc[r][1], acc[r][5]), s2 = __fadd_rn(acc[r][2], acc[r][6]), s3 = __fadd_rn(acc[r][3], acc[r][7]);
        const float res = __fadd_rn(__fadd_rn(s0, s2), __fadd_rn(s1, s3));
        const float mm = __fadd_rn(__fadd_rn(accm[r][0], accm[r][2]), __fadd_rn(accm[r][1], accm[r][3]));
        y[r] = __fadd_rn(res, mm);
    }
    if constexpr (EPI == 1) { // wave 1 (the up tile) hands its rows to wave 0 (the gate tile)
        if (wave == 1) *(float4 *)(xup + lane * 4) = make_float4(y[0], y[1], y[2], y[3]);
        __syncthreads();
        if (wave == 0 && col < p.bs) {
            const float4 up = *(const float4 *)(xup + lane * 4);
            const float uv[4] = {up.x, up.y, up.z, up.w};
            const int64_t row0 = (int64_t)task * 16 + kb * 4;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) // ps_silu_mul with the table in LDS
                o[r] = __fmul_rn(__fmul_rn(y[r], __fdiv_rn(1.0f, __fadd_rn(1.0f, ps_expf_glibc(-y[r], tab)))), uv[r]);
            *(float4 *)(W.out + (int64_t)col * W.ldo + row0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else if (col < p.bs) {
        const int64_t row0 = (int64_t)tile * 16 + kb * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = y[r];
            if (W.bias) v[r] = __fadd_rn(v[r], W.bias[row0 + r]);
            if (p.residual && wi == 0) v[r] = __fadd_rn(p.residual[(int64_t)col * W.ldo + row0 + r], v[r]);
        }
        g4k_store_pair<2>(p, wi, W, col, row0, v[0], v[1]);
        g4k_store_pair<2>(p, wi, W, col, row0 + 2, v[2], v[3]);
    }
}

// (a & mask) | magic as ONE instruction: two constants are two scalar operands, which a VOP3 instruction cannot read, so hipcc emits v_and + v_or;
// with the magic in a vector register it is v_and_or_b32
__device__ __forceinline__ uint32_t g4k_and_or(const uint32_t a, const uint32_t mask, const uint32_t magic_v) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(magic_v));
    return r;
}
// ---- narrow batches, few tiles, round 4: the K walk taken OFF the waves.  Both forms above walk a tile's super-blocks in order -- a
// step is ~0.43 us of one wave's instruction issue and barriers whatever was tried, 56 of them for the down projection (31 us, 1.1 TB/s).
// But the reference's order only binds the fp32 chains acc[u] = fma(d * yd, (float)sumi[u], acc[u]): the integer sums are exact in any
// order.  So the NWV waves of a workgroup take the super-blocks of a ROUND (sb = round * NWV + wave) side by side -- each one wave's
// work of gemm4k_wav_kernel for its super-block: all eight accumulator lanes and the four mins lanes through the matrix cores -- and
// park the sums with d * yd, -dmin * yd in LDS; after one barrier every wave runs the chains IT owns over the round's super-blocks
// in order (NWV = 8: accumulator lane u = wave and half a mins lane; NWV = 4: two lanes and a mins lane).
// 6 / 12 chains x NWV fmas per wave and round against ~250 instructions of operand building: the walk is now parallel over waves.
template <int NWV>
__global__ __launch_bounds__(NWV * 64) void gemm4k_par_kernel(const G4KParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t par_lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int m = lane & 15, kb = lane >> 4;
    float4 *const S = (float4 *)par_lds;  // [slot][u][lane]: (float)sumi[u] of rows 4 kb + r
    float4 *const M = S + NWV * 8 * 64;   // [slot][v][lane]: the mins products
    float4 *const D = M + NWV * 4 * 64;   // [slot][2][lane]: d * yd, -dmin * yd of rows 4 kb + r
    uint32_t *const trw = (uint32_t *)(D + NWV * 2 * 64) + wave * (16 * 36); // the wave's transposition buffer (gemm4k_wav_kernel)
    unsigned long long *const dbg = (PS_TL(p.dbg) && blockIdx.x < 1024 && lane == 0 && wave == 0) ? p.dbg + (size_t)blockIdx.x * 64 : nullptr; // timeline (tools/par_timeline.py)
    int dbg_n = 1;
    auto mark = [&]() { if (dbg && dbg_n < 29) dbg[dbg_n++] = __builtin_amdgcn_s_memtime(); };
    if (dbg) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[29] = __builtin_amdgcn_s_memrealtime(); }
    const int nsb = p.nsb;
    int wi, pair;
    const G4KRows R = g4k_rows<0>(p, (int)(blockIdx.x >> 1), wi, pair);
    const G4KMat &W = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : p.w[2]);
    const int tt = (int)(blockIdx.x & 1), tile = tt ? R.tile[1] : R.tile[0];
    const size_t g8 = (size_t)2 * tile + (m >> 3);
    const uint8_t *qb = R.qs[0] + ((size_t)2 * tile * nsb << 10) + (size_t)lane * 16;
    const uint8_t *hb = R.aux[0] + g8 * nsb * 128 + (size_t)(m & 7) * 16;
    const int col = m, colc = col < p.bs ? col : p.bs - 1;
    const char *qfp = (const char *)p.qf + lane * 16;
    const uint8_t *ydp = p.mf + colc * 4, *b16p = p.mf + 64 + colc * 32;

    // The wave's super-blocks of the next RA rounds (HBM), its B fragments one round ahead and the 16-sums two (L2).  Loads retire in
    // order: the far weight loads are the LAST thing a round issues, behind the fragment loads the next round waits for -- issued before
    // them they put a full memory latency in front of every round (RA = 2, weights first: 2.6 us per round, down 18 us).
    constexpr int RA = 4;
    ps_u32x4 rq[RA][2], rh[RA], rb[8], rs[2][2];
    float ryd[2];
    auto clampsb = [&](const int sb) { return sb < nsb ? sb : nsb - 1; };
    auto load_a = [&](const int s, const int sbx) {
        const int sb = clampsb(sbx);
        rq[s][0] = __builtin_nontemporal_load((const ps_u32x4 *)(qb + ((size_t)sb << 10)));
        rq[s][1] = __builtin_nontemporal_load((const ps_u32x4 *)(qb + ((size_t)(nsb + sb) << 10)));
        rh[s] = *(const ps_u32x4 *)(hb + (size_t)sb * 128);
    };
    auto load_m = [&](const int s, const int sbx) {
        const int sb = clampsb(sbx);
        ryd[s]   = *(const float *)(ydp + (size_t)sb * 576);
        rs[s][0] = *(const ps_u32x4 *)(b16p + (size_t)sb * 576);
        rs[s][1] = *(const ps_u32x4 *)(b16p + (size_t)sb * 576 + 16);
    };
#pragma unroll
    for (int s = 0; s < RA - 1; s++) load_a(s, wave + s * NWV);
    load_m(0, wave); load_m(1, wave + NWV);
#pragma unroll
    for (int u = 0; u < 8; u++) rb[u] = *(const ps_u32x4 *)(qfp + ((size_t)clampsb(wave) << 13) + u * 1024);
    load_a(RA - 1, wave + (RA - 1) * NWV); // (as a round leaves it: the farthest weights last)

    constexpr int UPW = 8 / NWV;              // accumulator lanes a wave chains
    constexpr int MR = NWV == 8 ? 2 : 4;      // rows of its mins lane
    const int mv = NWV == 8 ? wave >> 1 : wave, mr0 = NWV == 8 ? 2 * (wave & 1) : 0;
    float acc[UPW][4], accm[MR];
#pragma unroll
    for (int j = 0; j < UPW; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < MR; r++) accm[r] = 0.f;
    const g4k_f4 zf = {0.f, 0.f, 0.f, 0.f};
    const g4k_h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f}, km1024 = {(_Float16)-1024.f, (_Float16)-1024.f}, k16th = {(_Float16)0.0625f, (_Float16)0.0625f};
    uint32_t magic = 0x64006400u;
    asm volatile("" : "+v"(magic)); // (a vector register: see g4k_and_or)
    const uint32_t sel0 = 0x04000400u | ((uint32_t)(2 * (kb & 1)) * 0x00010001u), sel1 = sel0 + 0x00010001u;
    uint32_t msel[4];
#pragma unroll
    for (int e = 0; e < 4; e++) msel[e] = kb == 0 ? 0x04000400u | (uint32_t)(e * 0x00010001u) : 0x040C040Cu;
    const int n_rounds = (nsb + NWV - 1) / NWV;
#pragma clang loop unroll(disable)
    for (int rd0 = 0; rd0 < n_rounds; rd0 += RA) {
#pragma unroll
        for (int s = 0; s < RA; s++) {
            const int rd = rd0 + s;
            if (rd >= n_rounds) break; // (workgroup-uniform)
            const int sb = rd * NWV + wave; // (past the end in a ragged last round: the wave works on the last super-block again, nobody chains its slot)
            {
                const float yd = ryd[s & 1];
                const ps_u32x4 h = rh[s];
                const uint32_t scb = (kb & 2) ? ((h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4)) : (h.y & 0x3f3f3f3fu);
                const uint32_t t0 = __builtin_amdgcn_perm(0x64646464u, scb, sel0), t1 = __builtin_amdgcn_perm(0x64646464u, scb, sel1);
                g4k_h2 s0, s1;
                __builtin_memcpy(&s0, &t0, 4); __builtin_memcpy(&s1, &t1, 4);
                // the high nibbles stay where they are (bits 4..7 of each half: fp16 1024 + 16 n) and meet s / 16: (1024 + 16 n) s / 16 - 64 s = n s, exact
                s0 = s0 - k1024; s1 = (s1 - k1024) * k16th;
                const g4k_h2 n0 = s0 * km1024, n1 = s1 * km1024;
                float dr[4], dmn[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t hx = (uint32_t)__shfl((int)h.x, 4 * kb + r, 64);
                    dr[r]  = __fmul_rn(yd, ps_h2f((uint16_t)(hx & 0xffff)));
                    dmn[r] = __fmul_rn(-yd, ps_h2f((uint16_t)(hx >> 16)));
                }
                D[(wave * 2 + 0) * 64 + lane] = make_float4(dr[0], dr[1], dr[2], dr[3]);
                D[(wave * 2 + 1) * 64 + lane] = make_float4(dmn[0], dmn[1], dmn[2], dmn[3]);
                mark(); // 1: weights landed, header decoded
                *(ps_u32x4 *)(trw + (lane >> 3) * 36 + (lane & 7) * 4) = rq[s][0];
                *(ps_u32x4 *)(trw + (8 + (lane >> 3)) * 36 + (lane & 7) * 4) = rq[s][1];
                uint32_t wq[8];
#pragma unroll
                for (int u = 0; u < 8; u++) wq[u] = trw[m * 36 + u * 4 + kb];
                if (dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); mark(); } // 2: transposed
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t w = wq[u];
                    const uint32_t w8 = w >> 8;
                    const uint32_t tq[4] = {g4k_and_or(w, 0x000F000Fu, magic), g4k_and_or(w8, 0x000F000Fu, magic),
                                            g4k_and_or(w, 0x00F000F0u, magic), g4k_and_or(w8, 0x00F000F0u, magic)};
                    uint32_t o[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        g4k_h2 v;
                        __builtin_memcpy(&v, &tq[k], 4);
                        v = __builtin_elementwise_fma(v, k < 2 ? s0 : s1, k < 2 ? n0 : n1);
                        __builtin_memcpy(&o[k], &v, 4);
                    }
                    const ps_u32x4 ao = {o[0], o[1], o[2], o[3]};
                    g4k_h8 av, bv;
                    __builtin_memcpy(&av, &ao, 16); __builtin_memcpy(&bv, &rb[u], 16);
                    const g4k_f4 si = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, zf, 0, 0, 0);
                    rb[u] = *(const ps_u32x4 *)(qfp + ((size_t)clampsb(sb + NWV) << 13) + u * 1024);
                    S[(wave * 8 + u) * 64 + lane] = make_float4(si[0], si[1], si[2], si[3]);
                }
                mark(); // 3: accumulator lanes
                {
                    const uint32_t mn[2] = {h.z & 0x3f3f3f3fu, ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4)};
                    const uint32_t bw[8] = {rs[s & 1][0].x, rs[s & 1][0].y, rs[s & 1][0].z, rs[s & 1][0].w, rs[s & 1][1].x, rs[s & 1][1].y, rs[s & 1][1].z, rs[s & 1][1].w};
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        // (k-groups 1..3 select the constant byte 0x00 instead of the min: fp16 1024, i.e. a zero operand, without a select)
                        const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, mn[v >> 1], msel[(2 * v) & 3]);
                        const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, mn[v >> 1], msel[((2 * v) & 3) + 1]);
                        g4k_h2 h0, h1;
                        __builtin_memcpy(&h0, &p0, 4); __builtin_memcpy(&h1, &p1, 4);
                        h0 = h0 - k1024; h1 = h1 - k1024;
                        const g4k_h4 am = {h0[0], h0[1], h1[0], h1[1]};
                        g4k_h4 bm;
                        { const ps_u32x2 b2 = {bw[2 * v], bw[2 * v + 1]}; __builtin_memcpy(&bm, &b2, 8); }
                        const g4k_f4 pr = __builtin_amdgcn_mfma_f32_16x16x16f16(am, bm, zf, 0, 0, 0);
                        M[(wave * 4 + v) * 64 + lane] = make_float4(pr[0], pr[1], pr[2], pr[3]);
                    }
                }
                load_m(s & 1, sb + 2 * NWV);
                // (hipcc puts s_waitcnt vmcnt(0) in front of the first round of the unrolled body: nothing far may be outstanding there, so
                //  the last slot's next load goes out a round later, with slot 0's)
                if (s == 0) load_a(RA - 1, sb + (RA - 1) * NWV);
                if (s != RA - 1) load_a(s, sb + RA * NWV);
            }
            mark(); // 4: mins lanes
            __syncthreads();
            mark(); // 5: barrier
            // ---- the chains this wave owns, over the round's super-blocks in order
            const int cnt = nsb - rd * NWV < NWV ? nsb - rd * NWV : NWV;
            // (requesting the operands of several super-blocks ahead of the first chain step does not help: the phase is bound by LDS bandwidth --
            //  every wave reads d * yd again, 192 KB per round of eight -- not by round trips: 0.55 -> 0.60 us, profiles/r04_par_timeline.txt)
#pragma unroll
            for (int q = 0; q < NWV; q++) {
                if (q < cnt) {
                    const float4 d4 = D[(q * 2 + 0) * 64 + lane];
                    const float dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < UPW; j++) {
                        const float4 s4 = S[(q * 8 + wave * UPW + j) * 64 + lane];
                        const float sq[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[j][r] = __fmaf_rn(dq[r], sq[r], acc[j][r]);
                    }
                    const float *mp = (const float *)&M[(q * 4 + mv) * 64 + lane] + mr0, *mdp = (const float *)&D[(q * 2 + 1) * 64 + lane] + mr0;
#pragma unroll
                    for (int r = 0; r < MR; r++) accm[r] = __fmaf_rn(mdp[r], mp[r], accm[r]);
                }
            }
            mark(); // 6: chains
            __syncthreads();
            mark(); // 7: barrier
        }
    }
    // ---- the waves meet (the slots are free behind the last barrier): waves 0 and 1 finish rows 4 kb + 2 wave, + 1 in hsum_float_8's order
#pragma unroll
    for (int j = 0; j < UPW; j++) S[(wave * UPW + j) * 64 + lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
#pragma unroll
    for (int r = 0; r < MR; r++) ((float *)&M[mv * 64 + lane])[mr0 + r] = accm[r];
    __syncthreads();
    if (wave < 2 && col < p.bs) {
        float v[2];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = 2 * wave + rr;
            float au[8], mq[4];
#pragma unroll
            for (int u = 0; u < 8; u++) au[u] = ((const float *)&S[u * 64 + lane])[r];
#pragma unroll
            for (int q = 0; q < 4; q++) mq[q] = ((const float *)&M[q * 64 + lane])[r];
            const float s0 = __fadd_rn(au[0], au[4]), s1 = __fadd_rn(au[1], au[5]), s2 = __fadd_rn(au[2], au[6]), s3 = __fadd_rn(au[3], au[7]);
            const float res = __fadd_rn(__fadd_rn(s0, s2), __fadd_rn(s1, s3));
            const float mm = __fadd_rn(__fadd_rn(mq[0], mq[2]), __fadd_rn(mq[1], mq[3]));
            v[rr] = __fadd_rn(res, mm);
        }
        const int64_t row0 = (int64_t)tile * 16 + kb * 4 + 2 * wave;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            if (W.bias) v[rr] = __fadd_rn(v[rr], W.bias[row0 + rr]);
            if (p.residual && wi == 0) v[rr] = __fadd_rn(p.residual[(int64_t)col * W.ldo + row0 + rr], v[rr]);
        }
        g4k_store_pair<2>(p, wi, W, col, row0, v[0], v[1]);
    }
    if (dbg) { dbg[31] = __builtin_amdgcn_s_memtime(); dbg[30] = __builtin_amdgcn_s_memrealtime(); }
}
constexpr int g4k_par_lds(const int nwv) { return nwv * ((8 + 4 + 2) * 64 * 16 + 16 * 36 * 4); }

template <int EPI, int RA, int MB>
static void g4k_launch_wav(hipStream_t st, const G4KParams &p) {
    const unsigned grid = (unsigned)(EPI == 1 ? p.n_tasks : 2 * p.n_tasks);
    psk_note_kernel("gemm4k_wav_kernel<%d, %d, %d>", EPI, RA, MB);
    hipLaunchKernelGGL((gemm4k_wav_kernel<EPI, RA, MB>), dim3(grid), dim3(EPI == 1 ? 128 : 64), 0, st, p);
}

static int g4k_launch(hipStream_t st, int n_cu, G4KParams &p, const int epi, const int64_t bs) {
    const int n_ct = (int)((bs + 15) / 16);
    // (Measured, not kept.  (1) 64 rows x 32 columns per workgroup for batches of at most 32 columns -- every prepared weight
    // operand meeting every live column -- is SLOWER, 12 wide 6.3 vs 5.9 ms: with N = 4096 rows there are only 64 such
    // workgroups for 256 CUs, and a K walk is sequential by the reference's accumulation order; narrow batches are bound by
    // steps x step time.  (2) Letting the consumer waves past the batch skip their step does not shorten it: a consumer's
    // step is its own chain of LDS round trips (operands, d / dmin, mins), 0.83 us whether two or eight waves walk it.
    // (3) Unrolling the steps in pairs behind one barrier, so that the second step's LDS reads issue under the first one's
    // arithmetic, needs more than the 168 registers of a twelve-wave workgroup: 30 spills, 12.2 k tok/s.)
    // 8B tree forward, ms by width (profiles/r02_tree_forward_latency_8b.json): 2: 4.7, 8: 4.7, 12: 4.8, 16: 4.8 (narrow kernel),
    // 32: 5.9, 64: 6.3, 128: 8.4 (wide); the round-1 kernels that spread a row group's integer work over producer waves
    // (gemm8): 2: 4.9, 8: 5.0, 12: 6.1, 16: 6.2, 32: 8.8, 64: 12.8, 96: 19.9.  PS_GEMM4K_MIN_COLS moves the switch (default 2).
    if (bs < ps_gemm4k_min_cols() || p.nsb % 4) return -1;
    static const bool no_wav = getenv("PS_NO_GEMM4K_WAV") != nullptr;                                     // (A/B switches for measurements)
    static const int wav_cfg = getenv("PS_GEMM4K_WAV_CFG") ? atoi(getenv("PS_GEMM4K_WAV_CFG")) : 0;
    const int par_cfg = g_g4k_par; // 0: off, 4 / 8: that many waves per tile, else by tile count
    const int ctw = n_ct <= 1 ? 1 : 4; // column tiles per workgroup: the narrow kernel for at most 16 columns (its two-tile form, CT = 2, spills: 13.7 ms per 8B forward against 5.9 ms, not instantiated)
    p.n_cb = (n_ct + ctw - 1) / ctw;
    p.n_items = (p.n_tasks + 7) / 8 * 8 * p.n_cb;
    // persistent: one workgroup per CU walks the items w, w + n_wg, ... -- as long as that keeps its column block fixed
    // (the consumers prefetch the next item's first fragments with this item's column pointers)
    int n_wg = p.n_items;
    if (n_cu > 0 && n_wg > n_cu && n_cu % (8 * p.n_cb) == 0) n_wg = n_cu;
    // the XCD-aware item order (g4k_item): column blocks per XCD; only where its arithmetic holds, else the round-2 order
    p.cbx = 0;
    if (g_g4k_cbx > 0 && n_wg == 256 && p.n_items % 256 == 0 && (p.n_cb & (p.n_cb - 1)) == 0 && p.n_cb <= 8) {
        int cbx = g_g4k_cbx < p.n_cb ? g_g4k_cbx : p.n_cb;
        while (cbx & (cbx - 1)) cbx &= cbx - 1; // a power of two
        if (cbx < p.n_cb) p.cbx = cbx;          // (cbx == n_cb IS the round-2 order)
    }
    const dim3 grid((unsigned)n_wg), blk((G4K_NC + G4K_NP) * 64), blkn((G4K_NC + G4K_NPN) * 64);
    static_assert(G4K_RING == 4, "nsb % 4 == 0 is what the producers' ring is unrolled for");
    constexpr int LDS1 = G4K_XCH + 8 * 64 * 8 * 2 * 4 + PS_EXP2F_N * 8;
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) {
        (void)hipFuncSetAttribute((const void *)gemm4k_kernel<0, PS_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, G4K_LDS);
        (void)hipFuncSetAttribute((const void *)gemm4k_kernel<1, PS_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, G4K_LDS);
        (void)hipFuncSetAttribute((const void *)gemm4k_kernel<0, PS_Q5_K>, hipFuncAttributeMaxDynamicSharedMemorySize, G4K_LDS);
        (void)hipFuncSetAttribute((const void *)gemm4k_narrow_kernel<0, 1, PS_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
        (void)hipFuncSetAttribute((const void *)gemm4k_narrow_kernel<1, 1, PS_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
        (void)hipFuncSetAttribute((const void *)gemm4k_narrow_kernel<0, 1, PS_Q5_K>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
    }
    if (p.wt == PS_Q5_K) { // (single matrix, EPI 0)
        if (epi != 0) return -1;
        if (ctw == 1) { psk_note_kernel("gemm4k_narrow_kernel<0, 1, 13>"); hipLaunchKernelGGL((gemm4k_narrow_kernel<0, 1, PS_Q5_K>), grid, blkn, LDS1, st, p); }
        else { psk_note_kernel("gemm4k_kernel<0, 13>"); hipLaunchKernelGGL((gemm4k_kernel<0, PS_Q5_K>), grid, blk, G4K_LDS, st, p); }
    } else if (ctw == 1 && n_ct == 1 && !no_wav && epi != 1 && par_cfg != 0 && !(2 * p.n_tasks >= 4 * n_cu || wav_cfg == 1)) {
        // at most 16 columns, few row tiles (Q / K / V, O, down of a tree batch): the super-blocks of a round side by side, the chains behind them
        // (any nsb).  8B, 12 wide, us per launch, four waves walking K -> this: see profiles/r04_tree12_kernel_stats.txt
        static unsigned long long attrp = 0;
        if (ps_first_on_device(&attrp)) {
            (void)hipFuncSetAttribute((const void *)gemm4k_par_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, g4k_par_lds(8));
            (void)hipFuncSetAttribute((const void *)gemm4k_par_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, g4k_par_lds(4));
        }
        const int nwv = par_cfg == 4 || par_cfg == 8 ? par_cfg : (2 * p.n_tasks > n_cu ? 4 : 8); // more tiles than CUs: two workgroups of four waves per CU
        if (nwv == 8) { psk_note_kernel("gemm4k_par_kernel<8>"); hipLaunchKernelGGL((gemm4k_par_kernel<8>), dim3((unsigned)(2 * p.n_tasks)), dim3(512), g4k_par_lds(8), st, p); }
        else { psk_note_kernel("gemm4k_par_kernel<4>"); hipLaunchKernelGGL((gemm4k_par_kernel<4>), dim3((unsigned)(2 * p.n_tasks)), dim3(256), g4k_par_lds(4), st, p); }
    } else if (ctw == 1 && n_ct == 1 && !no_wav && p.nsb % 8 == 0) { // at most 16 columns: the wave-autonomous form
        // 8B tree forward 12 wide, us per launch, staged narrow kernel -> these (profiles/r03_tree12_kernel_stats*.txt): gate / up 39.6 -> 23.6,
        // Q / K / V and O 13 -> 11, down 31 -> 31, lm_head 151 -> 88; the forward 4.59 -> 3.68 ms.  Measured and not kept: the accumulator
        // lanes of a tile split over four unsynchronised waves with dword loads (every 128-byte line fetched from L2 eight times: 46 us
        // for gate / up); one wave per tile for the few-tile launches (24.9 us on average: one wave per four SIMDs walks K alone); two
        // super-blocks per barrier in the four-wave form (down 30 us, Q / K / V and O 14.7).  The four-wave form stays at ~0.43 us per
        // super-block step whatever was tried on it: its LDS operands fetched a step ahead (16.7 us average per launch, as before), the
        // header operands derived once by the parking wave instead of by all four (170 -> ~110 instructions per step: 16.6 us), a fifth
        // wave that owns the HBM stream with an eight-deep fragment ring in the others (22.9 us).
        if (epi == 1) g4k_launch_wav<1, 2, 2>(st, p);
        else if (2 * p.n_tasks >= 4 * n_cu || wav_cfg == 1) g4k_launch_wav<0, 2, 2>(st, p); // many tiles (lm_head): a wave per tile, occupancy hides the latency
        else g4k_launch_wav<0, 2, 2>(st, p); // few tiles with PS_GEMM4K_PAR=0 (A/B only; round 3's four-waves-per-tile kernel for this case was removed in round 5: gemm4k_par_kernel covers it)
    } else if (ctw == 1) {
        if (epi == 1) { psk_note_kernel("gemm4k_narrow_kernel<1, 1, 12>"); hipLaunchKernelGGL((gemm4k_narrow_kernel<1, 1, PS_Q4_K>), grid, blkn, LDS1, st, p); }
        else { psk_note_kernel("gemm4k_narrow_kernel<0, 1, 12>"); hipLaunchKernelGGL((gemm4k_narrow_kernel<0, 1, PS_Q4_K>), grid, blkn, LDS1, st, p); }
    } else if (epi == 1) { psk_note_kernel("gemm4k_kernel<1, 12>"); hipLaunchKernelGGL((gemm4k_kernel<1, PS_Q4_K>), grid, blk, G4K_LDS, st, p); }
    else { psk_note_kernel("gemm4k_kernel<0, 12>"); hipLaunchKernelGGL((gemm4k_kernel<0, PS_Q4_K>), grid, blk, G4K_LDS, st, p); }
    return 0;
}

// Q4_K batched mat-mul from fragment-major Q8_K activations (act.qf).  -1: not covered (the caller takes gemm8m).
int psk_gemm4k(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K, int64_t bs) {
    static const bool off = getenv("PS_NO_GEMM4K") != nullptr; // (A/B switch for measurements)
    if (off || a.pro != 0 || a.n_w < 1 || !act.qf || K % 256) return -1;
    if (a.rope && (a.n_w != 3 || a.silu_pair || a.residual || bs > 16)) return -1;
    G4KParams p{};
    int pairs_total = 0;
    for (int i = 0; i < a.n_w; i++) {
        if (a.w[i]->dtype != PS_Q4_K || a.w[i]->K != K || a.w[i]->N % 32 || a.ldo[i] % 4) return -1;
        p.w[i] = G4KMat{a.w[i]->qs, a.w[i]->aux, nullptr, a.out[i], a.bias[i], a.w[i]->N, a.ldo[i], (int)(a.w[i]->N / 16)};
        pairs_total += p.w[i].n_tiles / 2;
    }
    const int epi = a.silu_pair ? 1 : 0;
    if (epi == 1 && (a.n_w != 2 || a.w[0]->N != a.w[1]->N || a.ldo[0] != a.ldo[1])) return -1;
    p.n_w = a.n_w; p.nsb = (int)(K / 256); p.bs = (int)bs;
    p.n_tasks = epi == 1 ? p.w[0].n_tiles : pairs_total;
    p.residual = a.residual; p.qf = act.qf; p.mf = act.mf;
    p.dbg = psk_gemv_dbg_buf(12 + epi, epi ? 0 : (a.n_w == 3 ? 0 : (K <= 8192 ? 1 : 2))); // keys 48 QKV, 49 O, 50 down, 52 gate/up
    p.wt = PS_Q4_K;
    if (a.rope) { p.rope = *a.rope; p.rope_on = 1; }
    return g4k_launch(st, n_cu, p, epi, bs);
}

// Q5_K (single matrix, the Q5_K_M mix): the same kernels with the Q5_K producer.  -1: not covered.
int psk_gemm5k(hipStream_t st, int n_cu, const psk_gemv6_args &a, ps_act act, int64_t K, int64_t bs) {
    static const bool off = getenv("PS_NO_GEMM5K") != nullptr; // (A/B switch for measurements)
    const ps_weight *w = a.w;
    if (off || !w || w->dtype != PS_Q5_K || w->K != K || K % 1024 || w->N % 32 || a.ldo % 4 || !act.qf) return -1;
    G4KParams p{};
    p.w[0] = G4KMat{w->qs, w->sc, w->qh, a.out, a.bias, w->N, a.ldo, (int)(w->N / 16)};
    p.n_w = 1; p.nsb = (int)(K / 256); p.bs = (int)bs; p.n_tasks = p.w[0].n_tiles / 2;
    p.residual = a.residual; p.qf = act.qf; p.mf = act.mf; p.dbg = nullptr; p.wt = PS_Q5_K;
    return g4k_launch(st, n_cu, p, 0, bs);
}

// Q6_K batched mat-mul from the same fragment-major activations.  -1: not covered (the caller keeps the 8-column mat-vec launches).
int psk_gemm6k(hipStream_t st, int n_cu, const psk_gemv6_args &a, ps_act act, int64_t K, int64_t bs) {
    static const bool off = getenv("PS_NO_GEMM6K") != nullptr; // (A/B switch for measurements)
    const ps_weight *w = a.w;
    static const int64_t min_cols = [] { const char *e = getenv("PS_GEMM6K_MIN_COLS"); return e ? (int64_t)atoll(e) : (int64_t)9; }(); // (8B Q4_K_M tree forward, ms, threshold 17 / 9 / 2: 8 wide 5.9 / 6.0 / 6.2, 12: 6.9 / 6.3 / 6.3, 16: 7.6 / 6.4 / 6.3)
    if (off || !w || w->dtype != PS_Q6_K || w->K != K || K % 1024 || w->N % 32 || a.ldo % 2 || !act.qf || bs < min_cols || bs < ps_gemm4k_min_cols()) return -1;
    G6KParams p{};
    p.ql = w->qs; p.qh = w->qh; p.sc = (const int8_t *)w->sc; p.d = (const uint16_t *)w->aux;
    p.out = a.out; p.bias = a.bias; p.residual = a.residual; p.N = w->N; p.ldo = a.ldo;
    p.nsb = (int)(K / 256); p.bs = (int)bs; p.n_tasks = (int)(w->N / 32);
    p.qf = act.qf; p.mf = act.mf;
    const int n_ct = (int)((bs + 15) / 16);
    p.n_cb = (n_ct + 3) / 4;
    p.n_items = (p.n_tasks + 7) / 8 * 8 * p.n_cb;
    int n_wg = p.n_items;
    if (n_cu > 0 && n_wg > n_cu && n_cu % (8 * p.n_cb) == 0) n_wg = n_cu;
    static unsigned long long attr = 0;
    if (ps_first_on_device(&attr)) (void)hipFuncSetAttribute((const void *)gemm6k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G6K_LDS);
    psk_note_kernel("gemm6k_kernel");
    hipLaunchKernelGGL(gemm6k_kernel, dim3((unsigned)n_wg), dim3((G4K_NC + G4K_NP) * 64), G6K_LDS, st, p);
    return 0;
}

// Whether psk_gemm4k(a with a.rope set) will take a Q / K / V launch of `bs` columns AND do RoPE + the KV append in its epilogue
// (tree verify, prompt tails; mirrors the checks of psk_gemm4k and g4k_launch): 8B tree forward 12 wide 4.72 -> 4.62 ms.
bool psk_gemm4k_rope_ok(const psk_gemv_args &a, int64_t K, int64_t bs) {
    static const bool off = getenv("PS_NO_GEMM4K") != nullptr || getenv("PS_NO_ROPE_FUSION") != nullptr;
    if (off || a.n_w != 3 || a.silu_pair || a.residual || K % 1024 || bs < 2 || bs < ps_gemm4k_min_cols() || bs > 16) return false; // (at most 16 columns: the narrow kernel)
    for (int i = 0; i < 3; i++)
        if (!a.w[i] || a.w[i]->dtype != PS_Q4_K || a.w[i]->K != K || a.w[i]->N % 32 || a.ldo[i] % 4) return false;
    return true;
}
