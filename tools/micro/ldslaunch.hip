// Does a workgroup's LDS allocation cost launch time?  The same trivial kernel (every thread adds one LDS word to a global word) launched back to back on one
// stream, dependent (default barrier between launches), 256 workgroups, by threads per workgroup and dynamic LDS bytes.  Also: a streaming body (each workgroup
// reads 256 KiB) so that the boundary is the one between real mat-vec-like launches.
// build: hipcc --offload-arch=gfx950 -O3 -o ldslaunch ldslaunch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(float *out) {
    extern __shared__ float s[];
    if (threadIdx.x == 0) s[0] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += s[0];
}
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k_stream(const u4 *w, float *out, size_t per_wg) {
    extern __shared__ float s[];
    const u4 *p = w + (size_t)blockIdx.x * per_wg;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) { const u4 v = __builtin_nontemporal_load(p + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (threadIdx.x == 0) s[0] = (float)acc;
    __syncthreads();
    if (acc == 0x12345678u) out[1] = s[0];
}
int main() {
    float *out; hipMalloc(&out, 64); hipMemset(out, 0, 64);
    const size_t per_wg = 256 * 1024 / 16, layers = 16;
    u4 *w; hipMalloc(&w, layers * 256 * per_wg * 16); hipMemset(w, 1, layers * 256 * per_wg * 16);
    hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds_kb[] = {0, 16, 48, 64, 80, 96, 128, 157};
    for (int body = 0; body < 2; body++)
        for (int nt : {256, 576, 1024}) {
            printf("%s, %4d threads:", body ? "stream 64 MiB" : "empty        ", nt);
            for (int kb : lds_kb) {
                const int reps = 320;
                auto go = [&](int i) {
                    if (body) hipLaunchKernelGGL(k_stream, dim3(256), dim3(nt), (size_t)kb * 1024, 0, w + (size_t)(i % layers) * 256 * per_wg, out, per_wg);
                    else hipLaunchKernelGGL(k_empty, dim3(256), dim3(nt), (size_t)kb * 1024, 0, out);
                };
                for (int i = 0; i < 20; i++) go(i);
                hipEventRecord(e0, 0);
                for (int i = 0; i < reps; i++) go(i);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                printf("  %3d KiB %6.2f us", kb, 1e3 * ms / reps);
            }
            printf("\n");
        }
    return 0;
}
