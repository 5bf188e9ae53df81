// micro-benchmark: cost of a device-wide barrier among resident workgroups (atomic counter + spin), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// relaxed polling, one acquire fence at the end
__device__ __forceinline__ void grid_barrier2(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}
// two levels: 8 leaf counters (one per XCD-sized slice of the grid), the last arriver of a leaf bumps the root,
// the last arriver of the root publishes the generation; everybody polls the generation word
__device__ __forceinline__ void grid_barrier3(unsigned *ctr, unsigned gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned leaf = blockIdx.x & 7, per_leaf = gridDim.x >> 3;
        const unsigned a = __hip_atomic_fetch_add(ctr + 16 * (1 + leaf), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (a == gen * per_leaf - 1) {
            const unsigned b = __hip_atomic_fetch_add(ctr + 16 * 9, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (b == gen * 8 - 1) __hip_atomic_store(ctr, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(2);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

template <int V>
__global__ void k_bar(unsigned *ctr, int n, float *data, unsigned long long *t) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int i = 0; i < n; i++) {
        data[blockIdx.x * blockDim.x + threadIdx.x] = acc + i; // something to publish
        if (V == 1) grid_barrier(ctr, (unsigned)(i + 1) * gridDim.x);
        else if (V == 2) grid_barrier2(ctr, (unsigned)(i + 1) * gridDim.x);
        else grid_barrier3(ctr, (unsigned)(i + 1));
        acc += data[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x]; // ... and to consume
    }
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    if (acc == 12345.f) data[0] = acc;
}
__global__ void k_null() {}

template <int V>
void run(int n_cu) {
    for (int per_cu = 1; per_cu <= 2; per_cu++)
        for (int threads : {256, 512}) {
            const int grid = n_cu * per_cu, n = 200;
            unsigned *ctr; float *data; unsigned long long *t;
            hipMalloc(&ctr, 4096); hipMalloc(&data, (size_t)grid * threads * 4); hipMalloc(&t, grid * 8);
            hipMemset(ctr, 0, 4096);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_bar<V>, dim3(grid), dim3(threads), 0, 0, ctr, 1, data, t); // warm
            hipDeviceSynchronize(); hipMemset(ctr, 0, 4096);
            hipEventRecord(e0); hipLaunchKernelGGL(k_bar<V>, dim3(grid), dim3(threads), 0, 0, ctr, n, data, t); hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("variant %d grid %4d x %3d threads: %.3f us per barrier\n", V, grid, threads, ms * 1e3 / n);
            hipFree(ctr); hipFree(data); hipFree(t);
        }
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    run<1>(n_cu); run<2>(n_cu); run<3>(n_cu);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_null, dim3(512), dim3(512), 0, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_null, dim3(512), dim3(512), 0, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel launches, back to back: %.3f us each\n", ms * 1e3 / 200);
    return 0;
}
