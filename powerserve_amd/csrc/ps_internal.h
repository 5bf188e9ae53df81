// Host-side internal structures of libps_hip (not part of the C-ABI).
#pragma once
#include "../../include/ps_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

// Every device allocation of the library goes through these two.  PS_HIP_GUARD=1 (a debugging mode, tools/gpu_guard.sh): each allocation gets a
// virtual range of its own whose last mapped byte is the allocation's last (rounded up to hipMalloc's 256-byte alignment; PS_HIP_GUARD=16: to 16 bytes) and an unmapped stretch behind it, so that a kernel
// reading or writing past the end of a buffer faults instead of silently landing in a neighbour (the batch V.p kernel did, round 5).
hipError_t ps_dev_malloc(void **p, size_t bytes);
hipError_t ps_dev_free(void *p);

struct ps_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    std::string err;
    // scratch for op-level mul_mat activation quantization
    void *act_buf = nullptr;
    size_t act_cap = 0;
    // small device staging for host-provided pos/tokens
    int32_t *i32_buf = nullptr;
    size_t i32_cap = 0;
    uint8_t *u8_buf = nullptr;
    size_t u8_cap = 0;
    float *rope_buf = nullptr;
    size_t rope_cap = 0;
};

// Device layout of a quantized weight matrix [K, N]: a LANE-MAJOR repack of the GGUF blocks, private to the
// backend (every GGUF byte appears exactly once; zero padding only in a partial last row group / unit).
// Rows are taken in groups of RG; a "unit" is 1 KiB of quant data of one row group, laid out so that the
// 16 bytes at  lane*16  are what GPU lane (row r, AVX accumulator lane u) of k_gemv.hip consumes:
//   Q4_K (RG 8, unit = 1 super-block):  qs [group][sb][r][u][j=0..3][4 B] = bytes 32j+4u..32j+4u+3 of the
//                                       super-block (low nibbles: quad u of sub-block 2j, high: of 2j+1)
//                                       aux[group][sb][r] 16 B = {d, dmin, scales[12]}
//   Q8_0 (RG 8, unit = 4 blocks):       qs [group][b4][r][u][blk=0..3][4 B] = quants 4u..4u+3 of block 4*b4+blk
//                                       aux[group][b4][r][4] fp16 d
//   Q4_0 (RG 16, unit = 4 blocks):      qs [group][b4][r][u'=0..3][blk][4 B] = bytes 4u'..4u'+3 of the block
//                                       (low nibbles: quad u', high nibbles: quad u'+4);  aux as Q8_0
//   Q6_K : planes per row, lane-major inside a super-block (k_gemv6.hip: one wave per row, lane = (sb % 8, u)):
//          qs [N][K/256][u=0..7][16 B] = ql bytes {4u.., 32+4u.., 64+4u.., 96+4u..}   qh [N][K/256][u][8 B] = qh bytes
//          {4u.., 32+4u..}   sc [N][K/16] int8   aux [N][K/256] fp16 d
//   Q5_K : the same per-row planes: qs [N][K/256][u][16 B] = qs bytes {4u.., 32+4u.., 64+4u.., 96+4u..} (byte 32 jj + e:
//          element e of sub-vector 2 jj in the low nibble, of 2 jj + 1 in the high one)   qh [N][K/256][u][4 B] = qh bytes 4u..
//          (bit b = fifth bit of sub-vector b)   sc [N][K/256] 16 B = {d, dmin, scales[12]}
//   F32  : qs [N][K] float
struct ps_weight {
    int dtype;
    int64_t K, N;
    uint8_t *qs  = nullptr;
    uint8_t *aux = nullptr; // d (fp16) / hdr
    uint8_t *qh  = nullptr;
    uint8_t *sc  = nullptr;
    uint64_t gguf_bytes = 0;
};

// Activation quantized for the integer dot (structure of arrays):
//   qs   [rows][K] int8
//   d    [rows][K/blk] float   (Q8_0: the fp16-rounded scale widened back to fp32; Q8_K: fp32 scale)
//   bs16 [rows][K/16] int16    sums of 16 consecutive quants (== block_q8_K.bsums; also kept for Q8_0)
//   qf   Q8_K only, optional: a second copy of the quants AS FP16 (exact: |q| <= 127) in the fragment-major order of the
//        Q4_K batched mat-mul (k_gemm4k.hip): per (16 columns, super-block) 8 KiB  [u][lane = kb * 16 + column % 16][half][4]
//        = quants 4u + (0, 2, 1, 3) of sub-block 2 kb + half -- one 16-B B operand of v_mfma_f32_16x16x32_f16 per lane and u
struct ps_act {
    int8_t *qs;
    float *d;
    int16_t *bs16;
    _Float16 *qf;
    uint8_t *mf; // with qf: the column metadata tile-major, per (16 columns, super-block) 576 B = d[16] (fp32) then bsums[16][16]
                 // (int16): a wave of the Q4_K batched mat-mul reads its 16 columns' scale and sums from 5 cache lines, not 48
};

// hipFuncSetAttribute is per device: `mask` (one static per kernel instantiation) remembers the devices that have it
static inline bool ps_first_on_device(unsigned long long *mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (*mask & bit) return false;
    *mask |= bit;
    return true;
}

#define PS_CHECK(ctx, call)                                                                          \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

#define PS_FAIL(ctx, msg)                                                                            \
    do {                                                                                             \
        (ctx)->err = (msg);                                                                          \
        return 2;                                                                                    \
    } while (0)

// smallest batch the Q4_K chunk mat-mul (k_gemm4k.hip) takes; the quantizer writes the fragment-major copy from here on
static inline int64_t ps_gemm4k_min_cols() {
    static const int64_t v = [] { const char *e = getenv("PS_GEMM4K_MIN_COLS"); return e ? (int64_t)atoll(e) : (int64_t)2; }();
    return v;
}
static inline size_t ps_act_bytes(int64_t K, int64_t rows) {
    // qs + d (worst case blk 32) + bs16, each 256-B aligned
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    return al((size_t)K * rows) + al((size_t)(K / 32) * rows * 4) + al((size_t)(K / 16) * rows * 2) + al((size_t)K * ((rows + 15) / 16 * 16) * 2) + al((size_t)((rows + 15) / 16) * (K / 256 + 1) * 576);
}
static inline ps_act ps_act_carve(void *base, int64_t K, int64_t rows) {
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    ps_act a;
    char *p = (char *)base;
    a.qs    = (int8_t *)p;
    p += al((size_t)K * rows);
    a.d = (float *)p;
    p += al((size_t)(K / 32) * rows * 4);
    a.bs16 = (int16_t *)p;
    p += al((size_t)(K / 16) * rows * 2);
    a.qf = (_Float16 *)p;
    p += al((size_t)K * ((rows + 15) / 16 * 16) * 2);
    a.mf = (uint8_t *)p;
    return a;
}

// ---- fp16 prefill perf mode (perf16.hip): dense GEMMs on dequantized fp16 copies of the layer matrices, NOT bit-exact
struct psf16;
int psf16_create(ps_hip_ctx *c, psf16 **out); // the mode's handle (the GEMM is perf16.hip's own kernel: no library)
void psf16_destroy(psf16 *f);
int psf16_dequantize(ps_hip_ctx *c, const ps_weight *w, float *rows_buf, int32_t *ids_buf, int rows_cap, _Float16 *out); // out [N][K]
int psf16_gemm(ps_hip_ctx *c, psf16 *f, const _Float16 *W, int64_t N, int64_t K, const _Float16 *x, int bs, float *out, int64_t ldo, float beta);
int psf16_gemm_n(ps_hip_ctx *c, psf16 *f, int n_w, const _Float16 *const *W, const int64_t *N, int64_t K, const _Float16 *x, int bs, float *const *out, const int64_t *ldo, float beta); // up to three matrices sharing x, one launch
void psf16_rmsnorm_to_h(hipStream_t st, const float *x, const float *w, float eps, int64_t K, int bs, _Float16 *y);
void psf16_to_h(hipStream_t st, const float *x, int64_t n, _Float16 *y);
void psf16_silu_mul_to_h(hipStream_t st, const float *g, const float *u, int64_t n, _Float16 *y);
void psf16_add_bias(hipStream_t st, float *y, const float *b, int64_t N, int bs);

// ---- kernel launchers (defined in k_*.hip); all enqueue on `st`
// activation quantization.  mode: 0 plain, 1 rmsnorm(x, w, eps) first, 2 silu(x)*x2 first
void psk_quantize_act(hipStream_t st, int vdt, int mode, const float *x, const float *x2, const float *w, float eps,
                      int64_t K, int64_t rows, ps_act out);
void psk_pack_act_blocks(hipStream_t st, int vdt, ps_act in, int64_t K, int64_t rows, void *out_blocks);
void psk_repack_weight(hipStream_t st, int dtype, const uint8_t *raw, int64_t K, int64_t N, ps_weight *w);

// RoPE + KV-cache append fused into the QKV mat-vec epilogue (one activation column, adjacent-pair rotation)
struct psk_rope_kv {
    const struct ps_step_state *state; // pos0 = position of the token
    const float *rope_table;           // [n_ctx][head_size] (cos, sin) pairs
    float *k_cache, *v_cache;          // [n_ctx][kv_dim], [kv_dim][n_ctx]
    int head_size, n_dims, n_ctx, kv_dim;
    const int32_t *rope_pos;           // optional: RoPE position of the token (default: its cache slot)
    _Float16 *k16, *v16;               // optional fp16 mirrors, both [n_ctx][kv_dim] (fp16-KV decode mode, k_attn.hip)
};
struct psk_gemv_args {
    int n_w;                 // 1..3 matrices sharing the activation
    const ps_weight *w[3];
    float *out[3];           // [N_i][bs] row stride ldo[i] floats per batch column
    const float *bias[3];    // optional per-row bias (Qwen2)
    int64_t ldo[3];
    const float *residual;   // optional: out[0] = residual + y   (same layout as out[0])
    int silu_pair;           // 1: n_w==2, out[0][r] = silu(y0[r]) * y1[r]
    // activation source: pro 0 = `act` (already quantized), 1 = rmsnorm(pro_x, pro_norm_w, pro_eps) then
    // quantize, 2 = quantize(pro_x); pro_x is [bs][K] F32.  Done once per workgroup in the kernel prologue.
    int pro;
    const float *pro_x, *pro_norm_w;
    float pro_eps;
    // optional (n_w == 3, one column): out[0] receives the rotated q, k goes rotated to the K cache row pos0, v to the
    // V cache column pos0 (norm_attention.cpp:76-113); out[1] / out[2] are not written
    const psk_rope_kv *rope;
    int rope_wi0;            // with `rope`: which of Q / K / V w[0] is (0 .. 2) when the triple is split over launches by weight type
                             // (Q4_K_M: Q and K in Q4_K, V in Q6_K); n_w + rope_wi0 <= 3.  Taken by gemv4 / gemvk only.
};
bool psk_gemv_rope_ok(int wt, int64_t K); // the fused epilogue exists for this weight type / row length
size_t psk_gemv_lds_col_bytes(int wt, int64_t K);
int psk_gemm8(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K, int64_t bs); // -1: not covered
int psk_gemm4k(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K, int64_t bs);
bool psk_gemm4k_rope_ok(const psk_gemv_args &a, int64_t K, int64_t bs); // a Q / K / V batch whose RoPE + KV append the mat-mul epilogue will do (set a.rope) // Q4_K batches on v_mfma_f32_16x16x32_f16, exact integers (k_gemm4k.hip); -1: not covered
int psk_gemv_max_cols(int wt, int64_t K); // widest column group one launch takes (16, 8 or 4; > 4 needs pro == 0)
static inline int64_t ps_w_rg(int dtype) { return dtype == PS_Q4_0 ? 16 : 8; }
static inline int64_t ps_w_unit(int dtype) { return dtype == PS_Q4_K ? 256 : 128; }
// Q6_K / Q5_K x Q8_K (k_gemv6.hip): one matrix, any number of columns (groups of 8 inside); act must be quantized (Q8_K)
struct psk_gemv6_args {
    const ps_weight *w;
    float *out;            // [bs][ldo]
    int64_t ldo;
    const float *bias;     // optional [N]
    const float *residual; // optional, same layout as out (may alias out)
};
int psk_gemv6(hipStream_t st, int n_cu, const psk_gemv6_args &a, ps_act act, int64_t K, int64_t bs);
int psk_gemv5_multi(hipStream_t st, int n_cu, const psk_gemv6_args *a, int n_w, ps_act act, int64_t K, int64_t bs); // up to 3 Q5_K matrices, one launch
int psk_gemm5k(hipStream_t st, int n_cu, const psk_gemv6_args &a, ps_act act, int64_t K, int64_t bs); // Q5_K batches (k_gemm4k.hip); -1: not covered
int psk_gemm6k(hipStream_t st, int n_cu, const psk_gemv6_args &a, ps_act act, int64_t K, int64_t bs); // Q6_K, >= 17 columns (k_gemm4k.hip); -1: not covered
int psk_gemv_debug(int key, uint64_t *host_out, int n_words); // timeline buffer: arm / read back
int psk_gemv(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int vdt, int64_t K, int64_t bs);
// second-generation single-column Q4_K mat-vec (k_gemv4.hip); -1: not covered, the caller falls back
int psk_gemv4(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K);
bool psk_gemv4_covers(int64_t K);
// single-column Q4_0 / Q8_0 mat-vec, producer / chain-wave form (k_gemvb.hip); -1: not covered, the caller falls back
int psk_gemvb(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K);
bool psk_gemvb_covers(int wt, int64_t K);
// single-column Q6_K / Q5_K mat-vec with the fused prologues / epilogues (k_gemvk.hip); -1: not covered
int psk_gemvk(hipStream_t st, int n_cu, const psk_gemv_args &a, ps_act act, int64_t K);
bool psk_gemvk_covers(int wt, int64_t K);
unsigned long long *psk_gemv_dbg_buf(int epi, int pro); // timeline slot armed for this (epilogue, prologue) pair, or null
// the template instance the last quantized mat-vec / mat-mul launch of this process used (rocprofv3's kernel name): bench.py's
// roofline names the kernel that RAN, not the one it expects
void psk_note_kernel(const char *fmt, ...);
const char *psk_last_kernel();
