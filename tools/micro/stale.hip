// Does a relaxed agent-scope load (global_load ... sc1) ever return a line that this XCD's L2 fetched BEFORE another workgroup's
// write-through (sc1) store to it?  (attn_decode2's score exchange polls tagged granules; if the L2 could serve a stale copy
// for ever, a poll that started too early would never see the tag.)
//   reader: pre-reads the word (so that a caching L2 holds the line), tells the writer, waits for "written", then polls.
//   writer: waits for "pre-read done", stores the new value write-through, drains, says "written".
// Bounded loops; out[] says how far each side got and after how many polls / ticks the new value was seen.
// build: hipcc --offload-arch=gfx950 -O2 -o stale stale.hip ; run: ./stale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
template <int SYS> __device__ __forceinline__ u64 ld(const u64 *p) {
    return SYS ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int SYS> __device__ __forceinline__ void st(u64 *p, u64 v) {
    if (SYS) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned fl(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void fs(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

template <int SYS>
__global__ void k(u64 *data, unsigned *flags, long long *out, int wb, int rb, u64 nv, int n_words, int preread) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x == rb) {
        u64 pre = 0;
        if (preread) for (int w = 0; w < n_words; w++) pre += ld<SYS>(data + w * 16); // one word per 128-B line
        out[0] = (long long)pre;
        fs(flags + 0, 1u);
        int s = 0;
        while (fl(flags + 64) == 0 && s < 200000) { __builtin_amdgcn_s_sleep(4); s++; }
        out[1] = s;
        const long long t0 = __builtin_amdgcn_s_memtime();
        long long seen = -1, bad = 0;
        for (int it = 0; it < 20000 && seen < 0; it++) {
            bool all = true;
            for (int w = 0; w < n_words; w++) all = all && ld<SYS>(data + w * 16) == nv;
            if (all) seen = it; else { bad++; __builtin_amdgcn_s_sleep(2); }
        }
        out[2] = seen; out[3] = __builtin_amdgcn_s_memtime() - t0; out[4] = bad;
    } else if ((int)blockIdx.x == wb) {
        int s = 0;
        while (fl(flags + 0) == 0 && s < 200000) { __builtin_amdgcn_s_sleep(4); s++; }
        out[8] = s;
        for (int w = 0; w < n_words; w++) st<SYS>(data + w * 16, nv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fs(flags + 64, 1u);
    }
}
int main() {
    u64 *data; unsigned *flags; long long *out, h[16];
    hipMalloc(&data, 1 << 20); hipMalloc(&flags, 4096); hipMalloc(&out, 16 * 8);
    int nfail = 0;
    for (int sys = 0; sys < 2; sys++)
        for (int pre = 0; pre < 2; pre++)
            for (int pair = 0; pair < 3; pair++)
                for (int rep = 0; rep < 4; rep++) {
                    const int wb = 0, rb = pair == 0 ? 8 : (pair == 1 ? 1 : 5); // block b runs on XCD b % 8: pair 0 = same XCD
                    const int nw = rep < 2 ? 1 : 64;
                    hipMemset(data, 0, 1 << 20); hipMemset(flags, 0, 4096); hipMemset(out, 0xff, 16 * 8);
                    hipDeviceSynchronize();
                    const u64 nv = 0x1234567800000000ull + rep + 1;
                    if (sys) hipLaunchKernelGGL(k<1>, dim3(16), dim3(64), 0, 0, data, flags, out, wb, rb, nv, nw, pre);
                    else hipLaunchKernelGGL(k<0>, dim3(16), dim3(64), 0, 0, data, flags, out, wb, rb, nv, nw, pre);
                    hipDeviceSynchronize();
                    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                    printf("scope %s preread %d writer blk %d reader blk %d (%s XCD) words %2d: reader waited %lld polls for 'written'; new value seen at poll %lld (%lld bad polls, %lld ticks)%s\n",
                           sys ? "system" : "agent ", pre, wb, rb, pair == 0 ? "same" : "other", nw, h[1], h[2], h[4], h[3], h[2] < 0 ? "   <-- NEVER (stale)" : "");
                    if (h[2] < 0) nfail++;
                }
    printf("never-seen cases: %d\n", nfail);
    return 0;
}
