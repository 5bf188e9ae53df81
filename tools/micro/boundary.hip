// micro-benchmark: which kernel property makes a dependent kernel boundary expensive?  (graph replay, per-kernel time)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct Big { uint64_t a[50]; };

template <int VARIANT>
__global__ __launch_bounds__(1024) void k(const float *in, float *out, int n, Big b) {
    extern __shared__ float sm[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? in[i] : 0.f;
    if (VARIANT & 1) { // many live registers
        float r[96];
#pragma unroll
        for (int j = 0; j < 96; j++) r[j] = v * (float)(j + 1);
#pragma unroll
        for (int j = 0; j < 96; j++) v += r[j] * r[(j * 7) % 96];
    }
    if (VARIANT & 2) { // big code that is never executed (argument-dependent)
        if (b.a[7] == 12345) {
#pragma unroll
            for (int j = 0; j < 3000; j++) v = v * 1.0001f + (float)j;
        }
    }
    if (VARIANT & 4) { sm[threadIdx.x] = v; __syncthreads(); v = sm[threadIdx.x ^ 1]; }
    if (i < n) out[i] = v + 1.f + (float)(b.a[1] & 1);
}

template <int VARIANT>
static void run(const char *name, size_t lds) {
    hipStream_t st; hipStreamCreate(&st);
    float *a, *b; hipMalloc(&a, 1 << 22); hipMalloc(&b, 1 << 22); hipMemset(a, 0, 1 << 22); hipMemset(b, 0, 1 << 22);
    Big big{};
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *)k<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int n = 400;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k<VARIANT>, dim3(256), dim3(1024), lds, st, (i & 1) ? a : b, (i & 1) ? b : a, 262144, big);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %.2f us per kernel\n", name, ms * 1e3 / n);
}
int main() {
    run<0>("256 x 1024 threads, 400-byte kernarg, dependent r/w", 0);
    run<1>("+ ~100 live VGPRs", 0);
    run<2>("+ ~25 KB of (unexecuted) code", 0);
    run<4>("+ 70 KB dynamic LDS", 70 * 1024);
    run<7>("all three", 70 * 1024);
    return 0;
}
