// micro-benchmark: can two dependent weight-streaming kernels overlap when they are launched on two streams and the
// dependency (activation vector) is handed over in memory (arrival counter + cache-bypassing loads/stores)?
//   pattern 0: one stream, plain dependent launches (the kernel boundary is the dependency)
//   pattern 1: two streams, launches alternate; kernel i prefetches its first weights, then waits until kernel i-1's
//              workgroups have all arrived, then reads the activation with sc1 loads
//   pattern 2: pattern 1 captured into one hipGraph
// Each "layer" streams {14.2, 4.2(attn), 9.4, 66, 33} MB like the 8B Q4_K decode layer.  Also a census: how many CUs
// host two workgroups of the same launch (placement imbalance).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 4096;

struct Args {
    const uint8_t *w;      // this launch's weights
    uint32_t units_per_wg; // 1 KiB units per workgroup
    const float *xin;
    float *xout;
    unsigned *counter;     // arrival counter (monotonic inside one run)
    unsigned wait_target;  // arrivals of all earlier launches
    unsigned *census;      // [launch][256*?] optional
    int proto;             // 1: wait + coherent accesses + arrive
    unsigned *err;
    unsigned long long *tl; // [4] entry, wait done, prologue done, exit (block 0)
    int poll_sleep;
    unsigned long long *stamps; // [2][grid] entry / exit of every block
    int x_first;
};

__device__ __forceinline__ float coh_load(const float *p) { return __uint_as_float(__hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void coh_store(float *p, float v) { __hip_atomic_store((uint32_t *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NWAVE, int MINW>
__global__ __launch_bounds__(NWAVE * 64, MINW) void stream_k(const Args a) {
    __shared__ float xs[K];
    __shared__ float red[NWAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    float xpre[K / (NWAVE * 64)];
    if (a.x_first) {
#pragma unroll
        for (int i = 0; i < K / (NWAVE * 64); i++) xpre[i] = a.xin[threadIdx.x + i * NWAVE * 64];
    }
    const uint8_t *base = a.w + ((size_t)blockIdx.x * a.units_per_wg << 10);
    const uint32_t n = a.units_per_wg; // units of this workgroup, dealt round-robin to the waves in groups of 4
    // prefetch: two groups of four units per wave
    u32x4 qa[4], qb[4];
    uint32_t ua = wave * 4, ub = ua + NWAVE * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) qa[i] = __builtin_nontemporal_load((const u32x4 *)(base + ((size_t)min(ua + i, n - 1) << 10) + lane * 16));
#pragma unroll
    for (int i = 0; i < 4; i++) qb[i] = __builtin_nontemporal_load((const u32x4 *)(base + ((size_t)min(ub + i, n - 1) << 10) + lane * 16));
    if (a.census && threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        a.census[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 7) << 8); // xcc | se | sh+cu
    }
    if (a.tl && blockIdx.x == 0 && threadIdx.x == 0) a.tl[0] = __builtin_amdgcn_s_memrealtime();
    if (a.proto) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.wait_target) {
                for (int q = 0; q < a.poll_sleep; q++) __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 20)) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
    }
    if (a.tl && blockIdx.x == 0 && threadIdx.x == 0) a.tl[1] = __builtin_amdgcn_s_memrealtime();
    // "prologue": activation -> sum of squares -> scaled copy in LDS
    float ss = 0.f;
    if (a.x_first) {
#pragma unroll
        for (int i = 0; i < K / (NWAVE * 64); i++) { const float v = xpre[i]; xs[threadIdx.x + i * NWAVE * 64] = v; ss += v * v; }
    } else
    for (int i = threadIdx.x; i < K; i += NWAVE * 64) { const float v = a.proto ? coh_load(a.xin + i) : a.xin[i]; xs[i] = v; ss += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < NWAVE; i++) tot += red[i];
    const float sc = 1.0f / sqrtf(tot / K + 1e-5f);
    for (int i = threadIdx.x; i < K; i += NWAVE * 64) xs[i] *= sc;
    __syncthreads();
    if (a.tl && blockIdx.x == 0 && threadIdx.x == 0) a.tl[2] = __builtin_amdgcn_s_memrealtime();
    // main loop
    int acc = 0;
    const int *xi = (const int *)xs;
    for (uint32_t u0 = 0; u0 < n; u0 += 2 * NWAVE * 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = xi[((ua + i) * 64 + lane) & (K - 1)];
            acc = __builtin_amdgcn_sdot4((int)qa[i].x, y, acc, false); acc = __builtin_amdgcn_sdot4((int)qa[i].y, y, acc, false);
            acc = __builtin_amdgcn_sdot4((int)qa[i].z, y, acc, false); acc = __builtin_amdgcn_sdot4((int)qa[i].w, y, acc, false);
        }
        ua += 2 * NWAVE * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) qa[i] = __builtin_nontemporal_load((const u32x4 *)(base + ((size_t)min(ua + i, n - 1) << 10) + lane * 16));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = xi[((ub + i) * 64 + lane) & (K - 1)];
            acc = __builtin_amdgcn_sdot4((int)qb[i].x, y, acc, false); acc = __builtin_amdgcn_sdot4((int)qb[i].y, y, acc, false);
            acc = __builtin_amdgcn_sdot4((int)qb[i].z, y, acc, false); acc = __builtin_amdgcn_sdot4((int)qb[i].w, y, acc, false);
        }
        ub += 2 * NWAVE * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) qb[i] = __builtin_nontemporal_load((const u32x4 *)(base + ((size_t)min(ub + i, n - 1) << 10) + lane * 16));
    }
    // epilogue: K / 256 outputs per workgroup
    float r = (float)(acc & 0xff) * 1e-3f + 0.5f;
    const int oi = blockIdx.x * (K / 256) + (threadIdx.x & (K / 256 - 1));
    if (threadIdx.x < K / 256) { if (a.proto) coh_store(a.xout + oi, r); else a.xout[oi] = r; }
    if (a.proto) {
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.tl && blockIdx.x == 0 && threadIdx.x == 0) a.tl[3] = __builtin_amdgcn_s_memrealtime();
    if (a.stamps && threadIdx.x == 0) a.stamps[gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char **argv) {
    const int n_layers = 32;
    const double mb[5] = {14.2, 4.2, 9.4, 66.0, 33.0};
    size_t total = 0;
    std::vector<size_t> off, units;
    for (int L = 0; L < n_layers; L++) for (int j = 0; j < 5; j++) {
        size_t upw = (size_t)(mb[j] * 1e6 / 1024 / 256);
        upw = (upw + 63) / 64 * 64;
        off.push_back(total); units.push_back(upw);
        total += upw * 256 * 1024;
    }
    uint8_t *w; hipMalloc(&w, total); hipMemset(w, 0x11, total);
    float *x0, *x1; hipMalloc(&x0, K * 4); hipMalloc(&x1, K * 4); hipMemset(x0, 0, K * 4); hipMemset(x1, 0, K * 4);
    unsigned *ctr, *err; hipMalloc(&ctr, 256); hipMalloc(&err, 4);
    hipMemset(err, 0, 4);
    const int nk = (int)off.size();
    unsigned long long *stamps; hipMalloc(&stamps, (size_t)nk * 2 * 1024 * 8);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("total weights %.1f MB per pass, %d launches\n", total / 1e6, nk);
    auto run = [&](const char *name, auto kern, int threads, int wgs_per_cu, int x_first) {
        float best = 1e9f;
        const int grid = 256 * wgs_per_cu;
        for (int rep = 0; rep < 3; rep++) {
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int i = 0; i < nk; i++) {
                Args a{w + off[i], (uint32_t)(units[i] / wgs_per_cu), (i & 1) ? x1 : x0, (i & 1) ? x0 : x1, ctr, 0, nullptr, 0, err, nullptr, 1, rep == 2 ? stamps + (size_t)i * 2048 : nullptr, x_first};
                hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, st, a);
            }
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        std::vector<unsigned long long> t((size_t)nk * 2048); hipMemcpy(t.data(), stamps, t.size() * 8, hipMemcpyDeviceToHost);
        double dur[5] = {0}, bnd[5] = {0}, esk[5] = {0}, xsk[5] = {0}; int cnt[5] = {0};
        unsigned long long prev_last_exit = 0;
        for (int i = 0; i < nk; i++) {
            unsigned long long fe = ~0ull, le = 0, fx = ~0ull, lx = 0;
            for (int b = 0; b < grid; b++) { auto e = t[(size_t)i * 2048 + b], x = t[(size_t)i * 2048 + grid + b]; fe = std::min(fe, e); le = std::max(le, e); fx = std::min(fx, x); lx = std::max(lx, x); }
            if (i >= 5) { const int j = i % 5; dur[j] += (lx - fe) * 0.01; bnd[j] += ((double)fe - (double)prev_last_exit) * 0.01; esk[j] += (le - fe) * 0.01; xsk[j] += (lx - fx) * 0.01; cnt[j]++; }
            prev_last_exit = lx;
        }
        printf("%-44s %.3f ms = %.1f us/layer = %.2f TB/s\n", name, best, best * 1e3 / n_layers, total / (best * 1e-3) / 1e12);
        for (int j = 0; j < 5; j++) printf("      %5.1f MB: in-kernel %6.2f us (%.2f TB/s)  boundary before %5.2f  entry skew %5.2f  exit skew %5.2f\n", units[j] * 256 * 1024 / 1e6, dur[j] / cnt[j], units[j] * 256 * 1024 / (dur[j] / cnt[j]) / 1e6, bnd[j] / cnt[j], esk[j] / cnt[j], xsk[j] / cnt[j]);
    };
    run("1024 thr x 1/CU", stream_k<16, 4>, 1024, 1, 0);
    run("1024 thr x 1/CU, x first", stream_k<16, 4>, 1024, 1, 1);
    run("512 thr x 1/CU, x first", stream_k<8, 2>, 512, 1, 1);
    run("512 thr x 2/CU, x first", stream_k<8, 4>, 512, 2, 1);
    run("256 thr x 4/CU, x first", stream_k<4, 4>, 256, 4, 1);
    run("256 thr x 2/CU, x first", stream_k<4, 2>, 256, 2, 1);
    return 0;
}
