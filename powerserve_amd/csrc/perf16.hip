// fp16 prefill "perf mode" (second half; precedent: the reference's QNN graphs run fp16 activations,
// src/backend/qnn/causal_models.hpp:59-75) -- NOT bit-exact, opt-in (ps_hip_model_set_mode bit 5), never part of the headline.
// The quantized weights stay the source of truth (every parity path and every single-token step reads them); this mode adds a
// dequantized fp16 copy of the layer matrices and runs the mat-muls of a prefill chunk as plain dense GEMMs, fp16 x fp16 with fp32
// accumulation and fp32 output: no Q8_0 / Q8_K activation quantizer, no per-block fp32 chains -- the two things the reference's
// arithmetic costs on the matrix cores.  A dense GEMM is library work (rocBLAS, loaded with dlopen when the mode is first used: the
// backend has no link-time dependency on it, and without the library the mode reports an error instead of falling back); everything
// around it -- RMSNorm into fp16, SiLU(gate) * up into fp16, the conversions -- is the small kernels below, and RoPE, the KV append
// and the attention are the parity path's own kernels on the FP32 cache.
#include <dlfcn.h>
#include <rocblas/rocblas.h>
#undef rocblas_gemm_ex // (the header may alias it to the 64-bit entry; the symbol looked up below is the 32-bit one)

#include "ps_dev.h"
#include "ps_internal.h"
#include "ps_ops.h"

struct psf16 {
    void *lib = nullptr;
    rocblas_handle handle = nullptr;
    rocblas_status (*create)(rocblas_handle *) = nullptr;
    rocblas_status (*destroy)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*gemm_ex)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const void *, const void *,
                              rocblas_datatype, rocblas_int, const void *, rocblas_datatype, rocblas_int, const void *, const void *, rocblas_datatype,
                              rocblas_int, void *, rocblas_datatype, rocblas_int, rocblas_datatype, rocblas_gemm_algo, int32_t, uint32_t) = nullptr;
};

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
} // namespace

int psf16_create(ps_hip_ctx *c, psf16 **out) {
    psf16 *f = new psf16;
    for (const char *name : {"librocblas.so", "librocblas.so.5", "/opt/rocm/lib/librocblas.so"}) {
        f->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (f->lib) break;
    }
    if (!f->lib) { delete f; PS_FAIL(c, "fp16 perf mode: librocblas.so could not be loaded (the mode has no fallback)"); }
    f->create     = (decltype(f->create))dlsym(f->lib, "rocblas_create_handle");
    f->destroy    = (decltype(f->destroy))dlsym(f->lib, "rocblas_destroy_handle");
    f->set_stream = (decltype(f->set_stream))dlsym(f->lib, "rocblas_set_stream");
    f->gemm_ex    = (decltype(f->gemm_ex))dlsym(f->lib, "rocblas_gemm_ex");
    if (!f->create || !f->destroy || !f->set_stream || !f->gemm_ex) { dlclose(f->lib); delete f; PS_FAIL(c, "fp16 perf mode: rocBLAS entry points missing"); }
    if (f->create(&f->handle) != rocblas_status_success || f->set_stream(f->handle, c->stream) != rocblas_status_success) {
        dlclose(f->lib); delete f;
        PS_FAIL(c, "fp16 perf mode: rocblas_create_handle / rocblas_set_stream failed");
    }
    *out = f;
    return 0;
}
void psf16_destroy(psf16 *f) {
    if (!f) return;
    if (f->handle) (void)f->destroy(f->handle);
    // (the library stays mapped: unloading rocBLAS under a live HIP runtime is not worth the risk)
    delete f;
}

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

// out[bs][ldo] (fp32) = beta * out + x[bs][K] (fp16) . W[N][K]^T (fp16), fp32 accumulation.  Row-major operands are column-major
// transposes: out^T (N x bs) = W_cm^T (N x K) . x_cm (K x bs)
int psf16_gemm(ps_hip_ctx *c, psf16 *f, const _Float16 *W, int64_t N, int64_t K, const _Float16 *x, int bs, float *out, int64_t ldo, float beta) {
    const float alpha = 1.0f;
    const rocblas_status st = f->gemm_ex(f->handle, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int)N, (rocblas_int)bs, (rocblas_int)K, &alpha, W,
                                         rocblas_datatype_f16_r, (rocblas_int)K, x, rocblas_datatype_f16_r, (rocblas_int)K, &beta, out, rocblas_datatype_f32_r,
                                         (rocblas_int)ldo, out, rocblas_datatype_f32_r, (rocblas_int)ldo, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    if (st != rocblas_status_success) { c->err = "fp16 perf mode: rocblas_gemm_ex failed, status " + std::to_string((int)st); return 2; }
    return 0;
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
