// LDS-DMA semantics the decode mat-vec (k_gemv7.hip) relies on, checked on the device:
//   1. global_load_lds_dwordx4's instruction offset applies to the global AND the LDS address; lane l lands at M0 + offset + 16 l
//   2. M0 addresses the whole 160 KiB (destinations above 64 KiB)
//   3. an exec-masked request (lanes 0..31) writes only its own lanes' 16 bytes
//   4. a counted s_waitcnt vmcnt(N) retires the requests in issue order
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma ldsdma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p; }

__global__ __launch_bounds__(64) void k(const unsigned char *g, unsigned char *out, unsigned base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane16 = threadIdx.x * 16u;
    for (unsigned i = threadIdx.x; i < 160 * 1024 / 16; i += 64) ((uint4 *)smem)[i] = make_uint4(0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu);
    __syncthreads();
    const unsigned char *src = g + lane16, *hsrc = g + 8192 + (lane16 & 511u);
    const unsigned dst = lds_addr(smem) + base, hdst = dst + 4096;
    const unsigned long long low32 = 0xffffffffull;
    unsigned keep;
    unsigned long long keepx;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off nt\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %2, off offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %2, off offset:3072 nt\n\t"
                 "s_mov_b32 m0, %5\n\ts_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %6\n\ts_nop 1\n\t"
                 "global_load_lds_dwordx4 %3, off nt\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(keepx) : "v"(src), "v"(hsrc), "s"(dst), "s"(hdst), "s"(low32) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 160 * 1024 / 16; i += 64) ((uint4 *)out)[i] = ((uint4 *)smem)[i];
}

// in-order retirement: 16 requests of 1 KiB from widely spaced (cold) addresses; after vmcnt(16 - n) the first n slots must be complete
__global__ __launch_bounds__(64) void k_order(const unsigned char *g, unsigned *bad, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane16 = threadIdx.x * 16u;
    for (unsigned i = threadIdx.x; i < 16 * 1024 / 16; i += 64) ((uint4 *)smem)[i] = make_uint4(0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu);
    __syncthreads();
    const unsigned char *b = g + (size_t)blockIdx.x * 16 * stride;
    for (int i = 0; i < 16; i++) {
        const unsigned char *src = b + (size_t)i * stride + lane16;
        const unsigned dst = lds_addr(smem) + i * 1024;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    unsigned nb = 0;
    auto check = [&](int n) { // slots [0, n) must hold their data
        for (int i = 0; i < n; i++) {
            const uint4 v = *(const uint4 *)(smem + i * 1024 + lane16);
            const uint4 w = *(const uint4 *)(b + (size_t)i * stride + lane16);
            if (v.x != w.x || v.y != w.y || v.z != w.z || v.w != w.w) nb++;
        }
    };
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); check(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    check(16);
    if (nb) atomicAdd(bad, nb);
}

int main() {
    const size_t GB = 16 * 1024;
    std::vector<unsigned char> h(GB);
    for (size_t i = 0; i < GB; i++) h[i] = (unsigned char)((i * 7 + (i >> 8) * 13 + 1) & 0xff);
    unsigned char *g, *out;
    hipMalloc(&g, GB); hipMalloc(&out, 160 * 1024);
    hipMemcpy(g, h.data(), GB, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int rc = 0;
    for (unsigned base : {0u, 60u * 1024u, 150u * 1024u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, g, out, base);
        std::vector<unsigned char> o(160 * 1024);
        if (hipMemcpy(o.data(), out, o.size(), hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        size_t bad_units = 0, bad_hdr = 0, bad_rest = 0;
        for (size_t i = 0; i < o.size(); i++) {
            unsigned char want = 0xEE;
            if (i >= base && i < base + 4096) want = h[i - base];
            else if (i >= base + 4096 && i < base + 4096 + 512) want = h[8192 + (i - base - 4096)];
            if (o[i] != want) { if (i >= base && i < base + 4096) bad_units++; else if (i >= base + 4096 && i < base + 4608) bad_hdr++; else bad_rest++; }
        }
        printf("ldsdma base %6u: units %s (%zu bad bytes), masked header piece %s (%zu), nothing else touched %s (%zu)\n", base, bad_units ? "FAIL" : "ok", bad_units,
               bad_hdr ? "FAIL" : "ok", bad_hdr, bad_rest ? "FAIL" : "ok", bad_rest);
        rc |= (bad_units || bad_hdr || bad_rest);
    }
    // order
    const size_t stride = 1 << 16, blocks = 256;
    unsigned char *big; unsigned *bad;
    hipMalloc(&big, blocks * 16 * stride); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipMemset(big, 0x5a, blocks * 16 * stride);
    std::vector<unsigned char> pat(1024);
    for (size_t bidx = 0; bidx < blocks * 16; bidx++) { for (int j = 0; j < 1024; j++) pat[j] = (unsigned char)((bidx * 31 + j * 3 + 7) & 0xff); hipMemcpy(big + bidx * stride, pat.data(), 1024, hipMemcpyHostToDevice); }
    for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(k_order, dim3(blocks), dim3(64), 16 * 1024, 0, big, bad, stride);
    unsigned nb = 0;
    hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
    printf("ldsdma in-order retirement under counted vmcnt: %s (%u mismatching lane reads)\n", nb ? "FAIL" : "ok", nb);
    rc |= nb != 0;
    printf(rc ? "LDSDMA FAIL\n" : "LDSDMA PASS\n");
    return rc;
}
