// micro-benchmark: device-wide barrier with RELAXED agent-scope atomics only (no whole-cache release/acquire);
// the data crossing the barrier is written and read with agent-scope atomic (cache-bypassing) accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int HIER>
__device__ __forceinline__ void grid_barrier_relaxed(unsigned *ctr, unsigned gen) {
    __syncthreads(); // every thread's stores were issued ...
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0); // ... and thread 0's own are complete (vmcnt 0, lgkmcnt 0, expcnt 0)
        if (HIER) {
            const unsigned leaf = blockIdx.x & 7, per_leaf = gridDim.x >> 3;
            const unsigned a = __hip_atomic_fetch_add(ctr + 32 * (1 + leaf), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a == gen * per_leaf - 1) {
                const unsigned b = __hip_atomic_fetch_add(ctr + 32 * 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b == gen * 8 - 1) __hip_atomic_store(ctr, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
        } else {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * gridDim.x) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int HIER>
__global__ void k_bar(unsigned *ctr, int n, float *data, int *bad) {
    float acc = 0.f;
    const int me = blockIdx.x * blockDim.x + threadIdx.x, nb = ((blockIdx.x + 37) % gridDim.x) * blockDim.x + threadIdx.x;
    for (int i = 0; i < n; i++) {
        __hip_atomic_store(data + (i & 1) * gridDim.x * blockDim.x + me, (float)(i + 1) + (float)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0); // every thread waits for its own store before the workgroup barrier inside
        grid_barrier_relaxed<HIER>(ctr, (unsigned)(i + 1));
        const float v = __hip_atomic_load(data + (i & 1) * gridDim.x * blockDim.x + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (float)(i + 1) + (float)((blockIdx.x + 37) % gridDim.x)) atomicAdd(bad, 1);
        acc += v;
    }
    if (acc == 12345.f) data[0] = acc;
}

template <int HIER>
void run(int n_cu) {
    for (int per_cu = 1; per_cu <= 2; per_cu++)
        for (int threads : {256, 1024}) {
            if (per_cu == 2 && threads == 1024) continue;
            const int grid = n_cu * per_cu, n = 200;
            unsigned *ctr; float *data; int *bad;
            hipMalloc(&ctr, 4096); hipMalloc(&data, (size_t)2 * grid * threads * 4); hipMalloc(&bad, 4);
            hipMemset(ctr, 0, 4096); hipMemset(bad, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_bar<HIER>, dim3(grid), dim3(threads), 0, 0, ctr, 1, data, bad);
            hipDeviceSynchronize(); hipMemset(ctr, 0, 4096);
            hipEventRecord(e0); hipLaunchKernelGGL(k_bar<HIER>, dim3(grid), dim3(threads), 0, 0, ctr, n, data, bad); hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("relaxed %s grid %4d x %4d threads: %.3f us per barrier, stale reads %d\n", HIER ? "2-level" : "flat   ", grid, threads, ms * 1e3 / n, hb);
            hipFree(ctr); hipFree(data); hipFree(bad);
        }
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    run<0>(n_cu); run<1>(n_cu);
    return 0;
}
