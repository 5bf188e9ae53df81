// micro-benchmark: read bandwidth vs working-set size (is a second pass served by the memory-side cache?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(512) void rd(const u32x4 *p, size_t n16, uint32_t *sink) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = NT ? __builtin_nontemporal_load(p + i + k * stride) : p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}
int main() {
    const size_t MB = 1 << 20, maxb = 1024 * MB;
    u32x4 *buf; uint32_t *sink;
    hipMalloc(&buf, maxb); hipMalloc(&sink, 4); hipMemset(buf, 1, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; nt++)
        for (size_t mb : {16, 32, 64, 128, 192, 256, 384, 512, 1024}) {
            const size_t n16 = mb * MB / 16;
            for (int w = 0; w < 2; w++) { if (nt) hipLaunchKernelGGL(rd<1>, dim3(2048), dim3(512), 0, 0, buf, n16, sink); else hipLaunchKernelGGL(rd<0>, dim3(2048), dim3(512), 0, 0, buf, n16, sink); }
            hipDeviceSynchronize();
            const int reps = 10;
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) { if (nt) hipLaunchKernelGGL(rd<1>, dim3(2048), dim3(512), 0, 0, buf, n16, sink); else hipLaunchKernelGGL(rd<0>, dim3(2048), dim3(512), 0, 0, buf, n16, sink); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s working set %5zu MB: %7.2f TB/s (%.1f us per pass)\n", nt ? "nt   " : "plain", mb, (double)mb * MB * reps / (ms * 1e-3) / 1e12, ms * 1e3 / reps);
        }
    return 0;
}
