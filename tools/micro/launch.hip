// micro-benchmark: cost of a dependent kernel boundary in a stream and in a graph, for trivial kernels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct Big { uint64_t a[40]; };
__global__ void k_empty() {}
__global__ void k_arg(Big b, float *p) { if (b.a[3] == 77 && threadIdx.x == 0) p[0] = 1.f; }
__global__ void k_dep(const float *in, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.f;
}
__global__ void k_lds(const float *in, float *out, int n) {
    extern __shared__ float sm[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    sm[threadIdx.x] = i < n ? in[i] : 0.f;
    __syncthreads();
    if (i < n) out[i] = sm[threadIdx.x ^ 1] + 1.f;
}
template <typename F>
static void timeit(const char *name, F &&launch, hipStream_t st, int n = 400) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) launch(i);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < n; i++) launch(i);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s stream: %.2f us per kernel", name, ms * 1e3 / n);
    // the same sequence as a graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; i++) launch(i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("   graph: %.2f us per kernel\n", ms * 1e3 / n);
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    float *a, *b; hipMalloc(&a, 1 << 24); hipMalloc(&b, 1 << 24); hipMemset(a, 0, 1 << 24); hipMemset(b, 0, 1 << 24);
    Big big{}; big.a[3] = 1;
    timeit("empty <<<1,64>>>", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, st);
    timeit("empty <<<256,1024>>>", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, st); }, st);
    timeit("320-byte kernarg <<<256,1024>>>", [&](int) { hipLaunchKernelGGL(k_arg, dim3(256), dim3(1024), 0, st, big, a); }, st);
    timeit("dependent 16K floats ping-pong <<<64,256>>>", [&](int i) { if (i & 1) hipLaunchKernelGGL(k_dep, dim3(64), dim3(256), 0, st, a, b, 16384); else hipLaunchKernelGGL(k_dep, dim3(64), dim3(256), 0, st, b, a, 16384); }, st);
    timeit("dependent 256K floats, 64 KB LDS <<<256,1024>>>", [&](int i) { if (i & 1) hipLaunchKernelGGL(k_lds, dim3(256), dim3(1024), 65536, st, a, b, 262144); else hipLaunchKernelGGL(k_lds, dim3(256), dim3(1024), 65536, st, b, a, 262144); }, st);
    return 0;
}
