// What is the "first touch" at the head of a weight-streaming launch made of?  (a mat-vec launch whose first chunk was read shortly before starts 0.35-0.8 us sooner.)
// A probe launch (256 workgroups x 576 threads; every producer wave requests 4 KiB of ITS workgroup's share at entry, like gemv4's chunk 0, and stamps when they have landed, then
// streams the rest of a 37-KB share) runs on a region nobody has read for a long time, behind a 66-MB streaming launch, with between them:
//   0  nothing
//   1  a touch launch: one dword per 4 KiB page of every share (vector loads, workgroup b touches share b)
//   2  the same, one dword per 64 KiB
//   3  a touch launch reading EVERY line of the shares (the data is then in the memory-side cache)
//   4  one dword per 4 KiB by SCALAR loads (a different first-level translation path)
//   5  one dword per 4 KiB, but workgroup b touches share (b + 37) % 256 (another CU: only shared translation / cache levels can help)
//   6  every line, then 2 GB streamed past with non-temporal loads (what the mat-vecs use), then the probe: does the region survive a decode step's worth of streaming?
//   7  every line, then 2 GB streamed past with plain loads
//   8  every line, then 192 MB with non-temporal loads (less than the cache's size)
// build: hipcc --offload-arch=gfx950 -O3 -o firsttouch firsttouch.hip ; run: ./firsttouch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT>
__global__ __launch_bounds__(512) void stream_k(const u32x4 *p, size_t n16_per_wg, uint32_t *sink) {
    const u32x4 *q = p + (size_t)blockIdx.x * n16_per_wg;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < n16_per_wg; i += 512 * 4) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const u32x4 *a = q + (i + k * 512 < n16_per_wg ? i + k * 512 : i); v[k] = NT ? __builtin_nontemporal_load(a) : *a; }
#pragma unroll
        for (int k = 0; k < 4; k++) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}
__global__ __launch_bounds__(64) void touch_k(const uint8_t *y, size_t share, size_t gran, int shift, int scalar, uint32_t *sink) {
    const int b = ((int)blockIdx.x + shift) % (int)gridDim.x;
    const uint8_t *base = y + (size_t)b * share;
    uint32_t acc = 0;
    if (scalar) {
        if (threadIdx.x == 0)
            for (size_t o = 0; o < share; o += gran) { uint32_t v; asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base + o) : "memory"); acc ^= v; }
    } else {
        for (size_t o = (size_t)threadIdx.x * gran; o < share; o += 64 * gran) acc ^= *(const uint32_t *)(base + o);
    }
    if (acc == 0x12345678u) sink[1] = acc;
}
__global__ __launch_bounds__(576) void probe_k(const uint8_t *y, size_t share, unsigned long long *stamps, uint32_t *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const uint8_t *base = y + (size_t)blockIdx.x * share;
    u32x4 acc = {0, 0, 0, 0};
    if (wave < 8) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __builtin_nontemporal_load((const u32x4 *)(base + ((size_t)(wave * 4 + k) << 10) + lane * 16));
#pragma unroll
        for (int k = 0; k < 4; k++) acc ^= v[k];
        asm volatile("" : "+v"(acc)); // landed
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) stamps[(size_t)blockIdx.x * 16 + wave] = t1 - t0;
        for (size_t o = 32768 + (size_t)wave * 1024 + lane * 16; o + 16 <= share; o += 8192) acc ^= __builtin_nontemporal_load((const u32x4 *)(base + o));
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + 8] = t2 - t0;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[2] = acc.x;
}
int main() {
    const size_t MB = 1 << 20, X = 66 * MB, NWG = 256, share = 37 * 1024 + 512, region = NWG * share; // (shares are contiguous like a row-group split)
    uint8_t *buf; uint32_t *sink; unsigned long long *st;
    const size_t total = 6144 * MB;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&st, NWG * 16 * 8));
    CK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[9] = {"nothing", "touch 1 dword / 4 KiB", "touch 1 dword / 64 KiB", "touch every line", "scalar touch / 4 KiB", "touch / 4 KiB from another workgroup",
                             "every line, then 2 GB nt stream", "every line, then 2 GB plain stream", "every line, then 192 MB nt stream"};
    size_t cursor = 0;
    auto fresh = [&](size_t bytes) { if (cursor + bytes > total) cursor = 0; uint8_t *p = buf + cursor; cursor += (bytes + 2 * MB - 1) / (2 * MB) * (2 * MB); return p; };
    // age the whole buffer once
    for (size_t o = 0; o + X <= total; o += X) hipLaunchKernelGGL(stream_k<1>, dim3(NWG), dim3(512), 0, 0, (const u32x4 *)(buf + o), X / 16 / NWG, sink);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++)
        for (int var = 0; var < 9; var++) {
            uint8_t *x = fresh(X), *y = fresh(region);
            std::vector<unsigned long long> h(NWG * 16);
            hipLaunchKernelGGL(stream_k<1>, dim3(NWG), dim3(512), 0, 0, (const u32x4 *)x, X / 16 / NWG, sink);
            if (var == 1) hipLaunchKernelGGL(touch_k, dim3(NWG), dim3(64), 0, 0, y, share, (size_t)4096, 0, 0, sink);
            if (var == 2) hipLaunchKernelGGL(touch_k, dim3(NWG), dim3(64), 0, 0, y, share, (size_t)65536, 0, 0, sink);
            if (var == 3) hipLaunchKernelGGL(touch_k, dim3(NWG), dim3(64), 0, 0, y, share, (size_t)128, 0, 0, sink);
            if (var == 4) hipLaunchKernelGGL(touch_k, dim3(NWG), dim3(64), 0, 0, y, share, (size_t)4096, 0, 1, sink);
            if (var == 5) hipLaunchKernelGGL(touch_k, dim3(NWG), dim3(64), 0, 0, y, share, (size_t)4096, 37, 0, sink);
            if (var >= 6) {
                hipLaunchKernelGGL(touch_k, dim3(NWG), dim3(64), 0, 0, y, share, (size_t)128, 0, 0, sink);
                const size_t sz = var == 8 ? 192 * MB : 2048 * MB;
                if (cursor + sz > total) cursor = 0;
                uint8_t *z = buf + cursor; cursor += sz;
                if ((z <= y && y < z + sz)) { cursor = 0; z = buf; } // (keep the stream off the probed region)
                if (var == 7) hipLaunchKernelGGL(stream_k<0>, dim3(NWG), dim3(512), 0, 0, (const u32x4 *)z, sz / 16 / NWG, sink);
                else hipLaunchKernelGGL(stream_k<1>, dim3(NWG), dim3(512), 0, 0, (const u32x4 *)z, sz / 16 / NWG, sink);
            }
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe_k, dim3(NWG), dim3(576), 0, 0, y, share, st, sink);
            hipEventRecord(e1, 0);
            CK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            CK(hipMemcpy(h.data(), st, NWG * 16 * 8, hipMemcpyDeviceToHost));
            std::vector<double> land, life;
            for (size_t b = 0; b < NWG; b++) { for (int w = 0; w < 8; w++) land.push_back(h[b * 16 + w] / 100.0); life.push_back(h[b * 16 + 8] / 100.0); }
            std::sort(land.begin(), land.end()); std::sort(life.begin(), life.end());
            printf("rep %d  %-38s first 4 KiB per wave landed: p10 %5.2f p50 %5.2f p90 %5.2f max %5.2f us   workgroup lifetime p50 %5.2f max %5.2f us   launch (events) %6.2f us\n", rep, names[var],
                   land[land.size() / 10], land[land.size() / 2], land[land.size() * 9 / 10], land.back(), life[life.size() / 2], life.back(), ms * 1e3);
        }
    return 0;
}
