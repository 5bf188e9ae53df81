// fp16 prefill "perf mode" (second half; precedent: the reference's QNN graphs run fp16 activations,
// src/backend/qnn/causal_models.hpp:59-75) -- NOT bit-exact, opt-in (ps_hip_model_set_mode bit 5), never part of the headline.
// The quantized weights stay the source of truth (every parity path and every single-token step reads them); this mode adds a
// dequantized fp16 copy of the layer matrices and runs the mat-muls of a prefill chunk as dense GEMMs, fp16 x fp16 with fp32
// accumulation and fp32 output: no Q8_0 / Q8_K activation quantizer, no per-block fp32 chains -- the two things the reference's
// arithmetic costs on the matrix cores.  The GEMM is this file's own kernel (f16_gemm_kernel: v_mfma_f32_32x32x16_f16, 128 tokens x
// 128 / 64 weight rows per workgroup, K in blocks of 64 through a double-buffered, conflict-free LDS image; round 3 handed it to
// rocBLAS through dlopen); everything around it -- RMSNorm into fp16, SiLU(gate) * up into fp16, the conversions -- is the small
// kernels below, and RoPE, the KV append and the attention are the parity path's own kernels on the FP32 cache.
#include "ps_dev.h"
#include "ps_internal.h"
#include "ps_ops.h"

struct psf16 { int unused = 0; }; // (the mode's handle: nothing but a marker that the fp16 copies exist)

namespace {
__global__ void f32_to_f16_kernel(const float *x, _Float16 *y, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const float4 v = *(const float4 *)(x + i);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        *(h4 *)(y + i) = h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
}
// one workgroup per row: y = fp16(x * (w * rsqrt(mean(x^2) + eps)))  (the reference's RMSNorm, fp32 arithmetic, rounded to fp16 once)
__global__ __launch_bounds__(256) void rmsnorm_to_f16_kernel(const float *x, const float *w, float eps, int64_t K, _Float16 *y) {
    __shared__ float red[4];
    const float *xr = x + (int64_t)blockIdx.x * K;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < K; i += 256) s += xr[i] * xr[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float scale = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
    for (int64_t i = threadIdx.x; i < K; i += 256) y[(int64_t)blockIdx.x * K + i] = (_Float16)(xr[i] * (w[i] * scale));
}
__global__ void silu_mul_to_f16_kernel(const float *g, const float *u, _Float16 *y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gv = g[i];
        y[i] = (_Float16)(gv / (1.0f + __expf(-gv)) * u[i]);
    }
}
__global__ void add_bias_kernel(float *y, const float *b, int64_t N, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += b[i % N];
}
__global__ void iota_kernel(int32_t *p, int32_t first, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = first + i;
}
unsigned grid_for(int64_t n, int per = 256) { const int64_t g = (n + per - 1) / per; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// ---------------------------------------------------------------- the GEMM
// out[token][ldo] (fp32) = beta * out + x[token][K] (fp16) . W[row][K]^T (fp16) for up to three matrices that share x (Q / K / V, gate / up):
// both operands are K-contiguous, so the matrix instruction's A operand (32 rows x 16 k, lane l: row l % 32, k = 8 (l / 32) .. + 7 = one
// 16-byte piece) is a token tile and its B operand a weight-row tile; D[token][row] comes out with 32 consecutive weight rows of one token
// across lanes 0..31: the output stores are whole 128-byte lines.
//   * workgroup = 4 waves on 128 tokens x BN weight rows (BN 128: wave = 64 x 64 = 2 x 2 instructions' tiles; BN 64: 64 x 32), K walked in
//     blocks of 64: global -> registers (16-byte loads, the next block in flight while this one is multiplied) -> LDS rows of 128 + 16
//     bytes (36 dwords: the 16 rows a ds_read_b128 lane group touches land on 16 distinct bank quads) -> fragments; two LDS buffers,
//     one barrier per block;
//   * workgroup order: the token tiles of one weight tile sit 8 workgroups apart (same XCD, same time: the weight tile is fetched into
//     that L2 once), eight weight tiles per round;
//   * < 128 VGPRs, 74 KiB of LDS: two workgroups per CU.
typedef _Float16 f16_h8 __attribute__((ext_vector_type(8)));
typedef float f16_f16v __attribute__((ext_vector_type(16)));
struct F16Mat { const _Float16 *W; float *out; int64_t N, ldo; int tiles; };
struct F16Gemm { F16Mat w[3]; int n_w, bs, n_tm, tiles_total; const _Float16 *x; int64_t K; float beta; };
constexpr int F16_BM = 128, F16_BK = 64, F16_RS = F16_BK * 2 + 16; // tokens per workgroup, k per block, LDS row stride in bytes

template <int BN>
__global__ __launch_bounds__(256, 2) void f16_gemm_kernel(const F16Gemm p) {
    extern __shared__ __attribute__((aligned(16))) char f16_lds[];
    constexpr int A_BYTES = F16_BM * F16_RS, B_BYTES = BN * F16_RS, STAGE = A_BYTES + B_BYTES;
    constexpr int NB = BN / 64; // 16-byte B pieces per thread and k-block / 4;  instruction tiles per wave along the weight rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // (token tile, weight tile) of this workgroup
    const int per = 8 * p.n_tm, b_in = (int)blockIdx.x % per;
    const int tn_all = ((int)blockIdx.x / per) * 8 + (b_in & 7), tm = b_in >> 3;
    if (tn_all >= p.tiles_total) return;
    int wi = 0, tn = tn_all;
    if (p.n_w > 1 && tn >= p.w[0].tiles) { tn -= p.w[0].tiles; wi = 1; }
    if (p.n_w > 2 && wi == 1 && tn >= p.w[1].tiles) { tn -= p.w[1].tiles; wi = 2; }
    const _Float16 *W = wi == 0 ? p.w[0].W : (wi == 1 ? p.w[1].W : p.w[2].W);
    float *out        = wi == 0 ? p.w[0].out : (wi == 1 ? p.w[1].out : p.w[2].out);
    const int64_t ldo = wi == 0 ? p.w[0].ldo : (wi == 1 ? p.w[1].ldo : p.w[2].ldo);
    const int64_t K = p.K;
    const int t0 = tm * F16_BM, n0 = tn * BN;
    // loader roles: thread t fetches the 16-byte piece kq = t % 8 of rows t / 8 + 32 i
    const int lr = tid >> 3, kq = tid & 7;
    const _Float16 *ga[4], *gb[2 * NB];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int tok = t0 + lr + 32 * i;
        ga[i] = p.x + (int64_t)(tok < p.bs ? tok : 0) * K + kq * 8; // (tokens past the batch: row 0 once more, never stored)
    }
#pragma unroll
    for (int i = 0; i < 2 * NB; i++) gb[i] = W + (int64_t)(n0 + lr + 32 * i) * K + kq * 8;
    f16_h8 ra[4], rb[2 * NB];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; i++) ra[i] = *(const f16_h8 *)(ga[i] + (int64_t)kb * F16_BK);
#pragma unroll
        for (int i = 0; i < 2 * NB; i++) rb[i] = *(const f16_h8 *)(gb[i] + (int64_t)kb * F16_BK);
    };
    auto park = [&](int buf) {
        char *st = f16_lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++) *(f16_h8 *)(st + (lr + 32 * i) * F16_RS + kq * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < 2 * NB; i++) *(f16_h8 *)(st + A_BYTES + (lr + 32 * i) * F16_RS + kq * 16) = rb[i];
    };
    f16_f16v acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int nk = (int)(K / F16_BK);
    fetch(0);
    park(0);
    __syncthreads();
    const int fr = lane & 31, fk = (lane >> 5) * 16; // fragment row and byte offset of the lane's eight k inside a 16-k step
    for (int kb = 0; kb < nk; kb++) {
        if (kb + 1 < nk) fetch(kb + 1);
        const char *st = f16_lds + (kb & 1) * STAGE;
        const char *sa = st + (wm * 64 + fr) * F16_RS + fk, *sb = st + A_BYTES + (wn * (BN / 2) + fr) * F16_RS + fk;
#pragma unroll
        for (int ks = 0; ks < F16_BK / 16; ks++) {
            f16_h8 a[2], b[NB];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = *(const f16_h8 *)(sa + i * 32 * F16_RS + ks * 32);
#pragma unroll
            for (int j = 0; j < NB; j++) b[j] = *(const f16_h8 *)(sb + j * 32 * F16_RS + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kb + 1 < nk) park((kb + 1) & 1);
        __syncthreads();
    }
    // D register r of lane l: token row 8 (r / 4) + 4 (l / 32) + r % 4, weight row l % 32.  beta != 0 (the residual GEMMs): every old value is
    // requested before the first is used (a load behind a per-element condition is one L2 round trip per element: 64 in a row)
    float *const ob = out + n0 + wn * (BN / 2) + (lane & 31);
    const int tb = t0 + wm * 64 + 4 * (lane >> 5);
    if (p.beta != 0.f) { // (uniform)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NB; j++) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int tok = tb + i * 32 + 8 * (r >> 2) + (r & 3);
                    old[r] = ob[(int64_t)(tok < p.bs ? tok : 0) * ldo + j * 32];
                }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = __fmaf_rn(p.beta, old[r], acc[i][j][r]);
            }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tok = tb + i * 32 + 8 * (r >> 2) + (r & 3);
                if (tok < p.bs) ob[(int64_t)tok * ldo + j * 32] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------- the GEMM, large tiles (round 4)
// The kernel above moves 32 KiB from L2 and reads 48 KiB of fragments from LDS for every 128 x 128 x 64 block; at M = 512 its grids leave the
// chip half empty for the matrices of 4096 rows.  f16_gemm_w8_kernel: 256 tokens x 256 (or 128) weight rows per workgroup, eight waves (two per
// SIMD), a wave 128 x 64 (or 64 x 64) = 4 x 2 (2 x 2) instruction tiles -- half the bytes per flop on both paths.  The tiles go L2 -> LDS without
// touching registers (global_load_lds_dwordx4: lane l lands at M0 + 16 l): a piece = one wave-instruction = 8 rows x 128 B, and since the
// instruction fixes WHERE a lane's 16 bytes land but not WHICH 16 bytes it fetches, lane (row r, slot s) fetches piece s ^ (r / 2 % 8) of its
// row: the rows sit unpadded, 128 B apart, and a fragment read still meets every bank once (a ds_read_b128 is served in groups of 16 lanes --
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32 -- over 64 banks; the eight rows of a group that share row % 2 differ in row / 2 % 8).
// K in blocks of 64 (whole 128-byte lines per request), two stages; ONE barrier per block, placed between the third and the fourth group of
// matrix instructions: in front of it block k + 1 must have landed (s_waitcnt vmcnt(0): hipcc does not count these requests), behind it stage
// k is free, so the last group covers the first fragment reads of block k + 1 and the first requests of block k + 2.
// Measured and not kept (profiles/r04_f16_gemm.txt): the same tiles with FOUR waves, one per SIMD on 128 x 128 with 256 accumulator registers
// (K in blocks of 32 through a four-stage ring, reads and requests pinned one per gap between the matrix instructions): the same speed -- a
// request holds its wave's issue for 60-185 cycles (MI355X_MICROARCH.md) and alone on a SIMD nothing covers that; with the requests switched
// off the matrix instructions ran 639 ns per 32-k block, with everything on 780 (the eight-wave form: 770).  Both forms are bound by the
// request path: requests alone take as long as matrix instructions alone (431 vs 408 us for gate / up at 2048 tokens, 8.7 TB/s of L2 -> LDS).
__device__ __forceinline__ unsigned f16_lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p; }
__device__ __forceinline__ void f16_dma1(const char *q, const unsigned lds_dst) { // one wave-instruction: 1 KiB to lds_dst
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(q), "s"(lds_dst) : "memory");
}
template <int WMG, int MI, int NJ> // waves along the tokens (8 / WMG along the weight rows), instruction tiles per wave: MI x NJ
__global__ __launch_bounds__(512) void f16_gemm_w8_kernel(const F16Gemm p) {
    extern __shared__ __attribute__((aligned(16))) char f16_lds[];
    constexpr int WNG = 8 / WMG, BM = WMG * MI * 32, BN = WNG * NJ * 32, STAGE = (BM + BN) * 128, NI = (BM + BN) / 64; // NI: 8-row pieces per wave and k block
    static_assert(BM == 256 && NI % 3 != 1, "loader split");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WNG, wn = wave % WNG;
    const int per = 8 * p.n_tm, b_in = (int)blockIdx.x % per;
    const int tn_all = ((int)blockIdx.x / per) * 8 + (b_in & 7), tm = b_in >> 3;
    if (tn_all >= p.tiles_total) return;
    int wi = 0, tn = tn_all;
    if (p.n_w > 1 && tn >= p.w[0].tiles) { tn -= p.w[0].tiles; wi = 1; }
    if (p.n_w > 2 && wi == 1 && tn >= p.w[1].tiles) { tn -= p.w[1].tiles; wi = 2; }
    const _Float16 *W = wi == 0 ? p.w[0].W : (wi == 1 ? p.w[1].W : p.w[2].W);
    float *out        = wi == 0 ? p.w[0].out : (wi == 1 ? p.w[1].out : p.w[2].out);
    const int64_t ldo = wi == 0 ? p.w[0].ldo : (wi == 1 ? p.w[1].ldo : p.w[2].ldo);
    const int64_t K = p.K;
    const int t0 = tm * BM, n0 = tn * BN;
    // loader roles: the stage image is BM token rows, then BN weight rows, 128 B each; piece j of wave w = image rows 8 (NI w + j) .. + 7,
    // lane = (row lr, slot ls) fetching piece ls ^ (row / 2 % 8) of its row
    const int lr = lane >> 3, ls = lane & 7;
    const char *src[NI];
#pragma unroll
    for (int j = 0; j < NI; j++) {
        const int row = 8 * (NI * wave + j) + lr, kq = ls ^ ((row >> 1) & 7);
        if (row < BM) { const int tok = t0 + row; src[j] = (const char *)(p.x + (int64_t)(tok < p.bs ? tok : 0) * K) + kq * 16; } // (tokens past the batch: row 0 once more, never stored)
        else src[j] = (const char *)(W + (int64_t)(n0 + row - BM) * K) + kq * 16;
    }
    const unsigned lds0 = f16_lds_addr(f16_lds), my_dst = (unsigned)(NI * wave * 1024);
    f16_f16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int nk = (int)(K / 64);
    constexpr int R3 = NI / 3 + (NI % 3 == 2), R0 = NI / 3 + (NI % 3 == 2); // pieces requested in the groups 3 (of the block before) and 0; group 1 takes the rest
    auto request = [&](const int kb, const int cnt_lo, const int cnt_hi) { // pieces [cnt_lo, cnt_hi) of block kb (a block past the end: nothing)
        if (kb >= nk) return;
#pragma unroll
        for (int d = 0; d < NI; d++)
            if (d >= cnt_lo && d < cnt_hi) f16_dma1(src[d] + (int64_t)kb * 128, lds0 + (unsigned)((kb & 1) * STAGE) + my_dst + (unsigned)(d * 1024));
    };
    request(0, 0, NI);
    request(1, 0, R3);
    if (nk > 1) { if constexpr (R3 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5, fx = (fr >> 1) & 7;
    f16_h8 fa[2][MI], fb[2][NJ]; // fragments of k-step ks in set ks & 1
    auto frags = [&](const char *st, const int ks, f16_h8 (&a)[MI], f16_h8 (&b)[NJ]) {
        const int po = ((2 * ks + fh) ^ fx) << 4; // (the row bases are multiples of 32: (row / 2) % 8 = (fr / 2) % 8)
        const char *sa = st + (wm * MI * 32 + fr) * 128 + po, *sb = st + BM * 128 + (wn * NJ * 32 + fr) * 128 + po;
#pragma unroll
        for (int i = 0; i < MI; i++) a[i] = *(const f16_h8 *)(sa + i * 32 * 128);
#pragma unroll
        for (int j = 0; j < NJ; j++) b[j] = *(const f16_h8 *)(sb + j * 32 * 128);
    };
    auto mults = [&](const int set) {
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
    frags(f16_lds, 0, fa[0], fb[0]);
    for (int kb = 0; kb < nk; kb++) {
        const char *st = f16_lds + (kb & 1) * STAGE, *stn = f16_lds + ((kb + 1) & 1) * STAGE;
        frags(st, 1, fa[1], fb[1]);
        request(kb + 1, R3, R3 + R0);
        __builtin_amdgcn_sched_barrier(0);
        mults(0);
        __builtin_amdgcn_sched_barrier(0);
        frags(st, 2, fa[0], fb[0]);
        request(kb + 1, R3 + R0, NI);
        __builtin_amdgcn_sched_barrier(0);
        mults(1);
        __builtin_amdgcn_sched_barrier(0);
        frags(st, 3, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mults(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // block kb + 1 has landed (this wave's share)
        __syncthreads();
        if (kb + 1 < nk) frags(stn, 0, fa[0], fb[0]);
        request(kb + 2, 0, R3); // (stage kb & 1 is free behind the barrier)
        __builtin_amdgcn_sched_barrier(0);
        mults(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    float *const ob = out + n0 + wn * NJ * 32 + (lane & 31);
    const int tb = t0 + wm * MI * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            if (p.beta != 0.f) { // (uniform) every old value requested before the first is used
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int tok = tb + i * 32 + 8 * (r >> 2) + (r & 3);
                    old[r] = ob[(int64_t)(tok < p.bs ? tok : 0) * ldo + j * 32];
                }
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = __fmaf_rn(p.beta, old[r], acc[i][j][r]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int tok = tb + i * 32 + 8 * (r >> 2) + (r & 3);
                if (tok < p.bs) ob[(int64_t)tok * ldo + j * 32] = acc[i][j][r];
            }
        }
}

// reference for tools/f16_gemm_bench.py: one thread per output, fp32 accumulation in k order
__global__ void f16_gemm_ref_kernel(const _Float16 *x, const _Float16 *W, float *out, int M, int N, int64_t K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int t = (int)(i / N), n = (int)(i % N);
    float s = 0.f;
    for (int64_t k = 0; k < K; k++) s += (float)x[t * K + k] * (float)W[n * K + k];
    out[i] = s;
}
__global__ void f16_fill_kernel(_Float16 *p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
}
__global__ void f16_scale_kernel(float *p, int64_t n, float f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] *= f;
}
__global__ void f16_maxdiff_kernel(const float *a, const float *b, int64_t n, float *res) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned *)res, __float_as_uint(m)); // (non-negative floats order like their bit patterns)
}
} // namespace

int g_f16_variant = 0; // ps_hip_debug_set(4, v): 0 = by shape, 1 = the 128-token tiles, 2 / 3 = 256 tokens x 256 / 128 weight rows on the LDS-DMA path
int psf16_create(ps_hip_ctx *, psf16 **out) { *out = new psf16; return 0; }
void psf16_destroy(psf16 *f) { delete f; }

// fp16 copy [N][K] of a quantized weight: its rows through the backend's own dequantizer (get_rows, the embedding path), `rows_buf`
// = fp32 scratch for `rows_cap` rows, `ids_buf` = int32 scratch of the same count
int psf16_dequantize(ps_hip_ctx *c, const ps_weight *w, float *rows_buf, int32_t *ids_buf, int rows_cap, _Float16 *out) {
    for (int64_t r0 = 0; r0 < w->N; r0 += rows_cap) {
        const int n = (int)(w->N - r0 < rows_cap ? w->N - r0 : rows_cap);
        hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ids_buf, (int32_t)r0, n);
        psl_get_rows(c->stream, w, ids_buf, n, rows_buf);
        hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid_for((int64_t)n * w->K / 4)), dim3(256), 0, c->stream, rows_buf, out + r0 * w->K, (int64_t)n * w->K);
    }
    PS_CHECK(c, hipGetLastError());
    return 0;
}

// out_i[bs][ldo_i] (fp32) = beta * out_i + x[bs][K] (fp16) . W_i[N_i][K]^T (fp16), fp32 accumulation, for n_w <= 3 matrices sharing x in ONE launch
int psf16_gemm_n(ps_hip_ctx *c, psf16 *, int n_w, const _Float16 *const *W, const int64_t *N, int64_t K, const _Float16 *x, int bs, float *const *out, const int64_t *ldo, float beta) {
    if (n_w < 1 || n_w > 3 || bs < 1 || K % F16_BK != 0) PS_FAIL(c, "fp16 perf mode: GEMM shape not covered (K must be a multiple of 64)");
    bool wide = true;
    for (int i = 0; i < n_w; i++) { if (N[i] % 64 != 0) PS_FAIL(c, "fp16 perf mode: GEMM shape not covered (rows must be a multiple of 64)"); wide = wide && N[i] % 128 == 0; }
    F16Gemm p{};
    p.n_w = n_w; p.bs = bs; p.x = x; p.K = K; p.beta = beta;
    int64_t rows = 0;
    bool big = true, mid = true;
    for (int i = 0; i < n_w; i++) { rows += N[i]; big = big && N[i] % 256 == 0; mid = mid && N[i] % 128 == 0; }
    // 256-token tiles on the LDS-DMA path (f16_gemm_w8_kernel): 256 weight rows where that gives 70 % of the CUs a workgroup, else 128 where that gives
    // every CU one; variants 2 / 3 force them
    const int n_tm_big = (bs + 255) / 256;
    // (2048 tokens, us per launch, 128-token tiles -> these: gate / up 672 -> 553, Q / K / V 166 -> 116, O 90 -> 74, down 334 -> 288)
    int nj = 0;
    if (g_f16_variant == 2 || (g_f16_variant == 0 && big && rows / 256 * n_tm_big * 10 >= c->n_cu * 7)) nj = 4;
    else if (g_f16_variant == 3 || (g_f16_variant == 0 && mid && rows / 128 * n_tm_big >= c->n_cu)) nj = 2;
    if (nj) {
        const int bn = 64 * nj;
        for (int i = 0; i < n_w; i++) if (N[i] % bn) PS_FAIL(c, "fp16 perf mode: the large-tile GEMM needs rows in multiples of its tile");
        p.n_tm = n_tm_big;
        for (int i = 0; i < n_w; i++) { p.w[i] = F16Mat{W[i], out[i], N[i], ldo[i], (int)(N[i] / bn)}; p.tiles_total += p.w[i].tiles; }
        static unsigned long long attrd = 0;
        if (ps_first_on_device(&attrd)) {
            (void)hipFuncSetAttribute((const void *)f16_gemm_w8_kernel<2, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * 128);
            (void)hipFuncSetAttribute((const void *)f16_gemm_w8_kernel<4, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 128) * 128);
        }
        const dim3 grid((unsigned)((p.tiles_total + 7) / 8 * 8 * p.n_tm));
        if (nj == 4) hipLaunchKernelGGL((f16_gemm_w8_kernel<2, 4, 2>), grid, dim3(512), 2 * (256 + 256) * 128, c->stream, p);
        else hipLaunchKernelGGL((f16_gemm_w8_kernel<4, 2, 2>), grid, dim3(512), 2 * (256 + 128) * 128, c->stream, p);
        PS_CHECK(c, hipGetLastError());
        return 0;
    }
    p.n_tm = (bs + F16_BM - 1) / F16_BM;
    // 128-row weight tiles when that still gives every CU two workgroups, 64-row tiles otherwise
    const int bn = (wide && rows / 128 * p.n_tm >= 2 * c->n_cu) ? 128 : 64;
    for (int i = 0; i < n_w; i++) { p.w[i] = F16Mat{W[i], out[i], N[i], ldo[i], (int)(N[i] / bn)}; p.tiles_total += p.w[i].tiles; }
    const unsigned grid = (unsigned)((p.tiles_total + 7) / 8 * 8 * p.n_tm);
    const size_t smem = (size_t)2 * (F16_BM + bn) * F16_RS;
    static unsigned long long attr = 0; // devices that have the attribute
    if (ps_first_on_device(&attr)) {
        (void)hipFuncSetAttribute((const void *)f16_gemm_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (F16_BM + 128) * F16_RS);
        (void)hipFuncSetAttribute((const void *)f16_gemm_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (F16_BM + 64) * F16_RS);
    }
    if (bn == 128) hipLaunchKernelGGL(f16_gemm_kernel<128>, dim3(grid), dim3(256), smem, c->stream, p);
    else hipLaunchKernelGGL(f16_gemm_kernel<64>, dim3(grid), dim3(256), smem, c->stream, p);
    PS_CHECK(c, hipGetLastError());
    return 0;
}
int psf16_gemm(ps_hip_ctx *c, psf16 *f, const _Float16 *W, int64_t N, int64_t K, const _Float16 *x, int bs, float *out, int64_t ldo, float beta) {
    return psf16_gemm_n(c, f, 1, &W, &N, K, x, bs, &out, &ldo, beta);
}

void psf16_rmsnorm_to_h(hipStream_t st, const float *x, const float *w, float eps, int64_t K, int bs, _Float16 *y) {
    hipLaunchKernelGGL(rmsnorm_to_f16_kernel, dim3((unsigned)bs), dim3(256), 0, st, x, w, eps, K, y);
}
void psf16_to_h(hipStream_t st, const float *x, int64_t n, _Float16 *y) { // n % 4 == 0
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, x, y, n);
}
void psf16_silu_mul_to_h(hipStream_t st, const float *g, const float *u, int64_t n, _Float16 *y) {
    hipLaunchKernelGGL(silu_mul_to_f16_kernel, dim3(grid_for(n)), dim3(256), 0, st, g, u, y, n);
}
void psf16_add_bias(hipStream_t st, float *y, const float *b, int64_t N, int bs) {
    hipLaunchKernelGGL(add_bias_kernel, dim3(grid_for(N * bs)), dim3(256), 0, st, y, b, N, N * bs);
}

// tools/f16_gemm_bench.py: time one GEMM shape of the perf mode on synthetic operands and compare it with a k-ordered fp32 reference
extern "C" int ps_hip_debug_f16_gemm(ps_hip_ctx *c, int M, int64_t N, int64_t K, int reps, float beta, double *us_per_launch, double *max_abs_err) {
    if (!c || M < 1 || N < 1 || K < 1 || reps < 1) return 1;
    PS_CHECK(c, hipSetDevice(c->device));
    _Float16 *x = nullptr, *W = nullptr;
    float *o = nullptr, *r = nullptr, *res = nullptr;
    PS_CHECK(c, ps_dev_malloc((void **)&x, (size_t)M * K * 2));
    PS_CHECK(c, ps_dev_malloc((void **)&W, (size_t)N * K * 2));
    PS_CHECK(c, ps_dev_malloc((void **)&o, (size_t)M * N * 4));
    PS_CHECK(c, ps_dev_malloc((void **)&r, (size_t)M * N * 4));
    PS_CHECK(c, ps_dev_malloc((void **)&res, 4));
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(f16_fill_kernel, dim3(1024), dim3(256), 0, st, x, (int64_t)M * K, 17u);
    hipLaunchKernelGGL(f16_fill_kernel, dim3(1024), dim3(256), 0, st, W, (int64_t)N * K, 91u);
    PS_CHECK(c, hipMemsetAsync(o, 0, (size_t)M * N * 4, st));
    PS_CHECK(c, hipMemsetAsync(res, 0, 4, st));
    hipLaunchKernelGGL(f16_gemm_ref_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, st, x, W, r, M, (int)N, K);
    int rc = psf16_gemm(c, nullptr, W, N, K, x, M, o, N, 0.f);
    if (!rc && beta != 0.f) { // out = beta out + ...: once more on top of the first result, against (1 + beta) x the reference
        rc = psf16_gemm(c, nullptr, W, N, K, x, M, o, N, beta);
        hipLaunchKernelGGL(f16_scale_kernel, dim3(1024), dim3(256), 0, st, r, (int64_t)M * N, 1.f + beta);
    }
    if (!rc) {
        hipLaunchKernelGGL(f16_maxdiff_kernel, dim3(1024), dim3(256), 0, st, o, r, (int64_t)M * N, res);
        float h = 0.f;
        PS_CHECK(c, hipMemcpyAsync(&h, res, 4, hipMemcpyDeviceToHost, st));
        PS_CHECK(c, hipStreamSynchronize(st));
        *max_abs_err = h;
        hipEvent_t e0, e1;
        PS_CHECK(c, hipEventCreate(&e0)); PS_CHECK(c, hipEventCreate(&e1));
        for (int i = 0; i < 3 && !rc; i++) rc = psf16_gemm(c, nullptr, W, N, K, x, M, o, N, beta);
        PS_CHECK(c, hipEventRecord(e0, st));
        for (int i = 0; i < reps && !rc; i++) rc = psf16_gemm(c, nullptr, W, N, K, x, M, o, N, beta);
        PS_CHECK(c, hipEventRecord(e1, st));
        PS_CHECK(c, hipEventSynchronize(e1));
        float ms = 0.f;
        PS_CHECK(c, hipEventElapsedTime(&ms, e0, e1));
        *us_per_launch = 1e3 * ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)ps_dev_free(x); (void)ps_dev_free(W); (void)ps_dev_free(o); (void)ps_dev_free(r); (void)ps_dev_free(res);
    return rc;
}
