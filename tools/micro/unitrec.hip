// micro-benchmark: how many SIMD cycles does the integer work of one Q4_K unit (unit_rec, ps_gemv_dev.h) cost when
// nothing waits for memory?  Waves loop over register-resident "weights" against an LDS activation image and drop the
// records in LDS, as the producers of gemv3 / gemv4 do.  Reports shader cycles per unit and per SIMD at 1, 2, 3 and 4
// waves per SIMD (one workgroup per CU).
#include "../../powerserve_amd/csrc/ps_gemv_dev.h"
#include <cstdio>

template <int NWAVE, int VARIANT>
__global__ __launch_bounds__(NWAVE * 64) void k(const uint4 *w, int iters, unsigned long long *out, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int8_t *lq = (int8_t *)smem;            // 4096 quants
    float *ld  = (float *)(smem + 4096);    // 16 scales
    int *lb    = (int *)(ld + 16);          // 128 sums of 32
    int2 *recs = (int2 *)(smem + 8192);     // [NWAVE * 4][64]  (VARIANT >= 1: float4 records)
    float4 *recs4 = (float4 *)(smem + 8192);
    for (int i = threadIdx.x; i < 1024; i += NWAVE * 64) ((int *)lq)[i] = i * 0x01030507;
    for (int i = threadIdx.x; i < 128; i += NWAVE * 64) lb[i] = i;
    __syncthreads();
    LAct A; A.q32 = (const int *)lq; A.d = ld; A.bs32 = lb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane & 7;
    uint4 q[4], h[4];
    for (int i = 0; i < 4; i++) { q[i] = w[(wave * 4 + i) * 64 + lane]; h[i] = w[4096 + i * 8 + (lane >> 3)]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ul = (it * 4 + i) & 15;
            int2 rc = unit_rec<PS_Q4_K>(q[i], h[i], ul, u, A);
            if (VARIANT == 0) recs[(wave * 4 + i) * 64 + lane] = rc;
            else {
                const float yd = A.d[ul];
                const float d    = __fmul_rn(yd, ps_h2f((uint16_t)(h[i].x & 0xffff)));
                const float dmin = __fmul_rn(-yd, ps_h2f((uint16_t)(h[i].x >> 16)));
                recs4[(wave * 4 + i) * 64 + lane] = make_float4(d, (float)rc.x, dmin, (float)rc.y);
            }
            q[i].x ^= (unsigned)rc.x; // the next trip depends on this one: nothing is hoisted out of the loop
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * NWAVE + wave] = t1 - t0;
    if (q[0].x == 0x12345) sink[0] = 1;
}

template <int NWAVE, int VARIANT = 0>
static void run(const uint4 *w, unsigned long long *out, int *sink) {
    const int iters = 2000;
    hipFuncSetAttribute((const void *)k<NWAVE, VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((k<NWAVE, VARIANT>), dim3(256), dim3(NWAVE * 64), 8192 + NWAVE * 4 * 64 * 16, 0, w, iters, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * NWAVE);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per_wave_unit = s / h.size() / (iters * 4.0);
    printf("variant %d %2d waves per CU (%.2f per SIMD): %7.1f cycles per unit per wave, %7.1f SIMD cycles per unit\n", NWAVE, NWAVE / 4.0, per_wave_unit, per_wave_unit / (NWAVE / 4.0));
}

int main() {
    uint4 *w; hipMalloc(&w, (4096 + 64) * 16); hipMemset(w, 0x35, (4096 + 64) * 16);
    unsigned long long *out; hipMalloc(&out, 256 * 16 * 8);
    int *sink; hipMalloc(&sink, 4);
    run<4>(w, out, sink); run<8>(w, out, sink); run<12>(w, out, sink); run<16>(w, out, sink);
    run<4, 1>(w, out, sink); run<8, 1>(w, out, sink); run<12, 1>(w, out, sink); run<16, 1>(w, out, sink);
    return 0;
}
