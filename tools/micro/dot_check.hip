#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../powerserve_amd/csrc/ps_dev.h"
__global__ void k(const uint32_t *in, int *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = in[4 * i], yl = in[4 * i + 1], yh = in[4 * i + 2], scp = in[4 * i + 3] & 0x3f3f3f3fu;
    const uint32_t M = 0x0f0f0f0fu;
    const int dl0 = dot4((int)(w & M), (int)yl, 0), dh0 = dot4((int)((w >> 4) & M), (int)yh, 0);
    const int ref = (int)(scp & 0xff) * dl0 + (int)((scp >> 8) & 0xff) * dh0 + 5;
    const uint32_t sc16 = __builtin_amdgcn_perm(0u, scp, 0x0c010c00u);
    const int dl = dot4z((int)(w & M), (int)yl), dh = dot4z((int)((w >> 4) & M), (int)yh);
    const uint32_t d16 = __builtin_amdgcn_perm((uint32_t)dh, (uint32_t)dl, 0x05040100u);
    out[8 * i + 0] = ref; out[8 * i + 1] = dot2_i16(d16, sc16, 5);
    out[8 * i + 2] = dl0; out[8 * i + 3] = dl; out[8 * i + 4] = dh0; out[8 * i + 5] = dh; out[8 * i + 6] = (int)sc16; out[8 * i + 7] = (int)d16;
}
int main() {
    const int n = 256;
    uint32_t h[4 * n]; for (int i = 0; i < 4 * n; i++) h[i] = (uint32_t)rand() * 2654435761u + rand();
    uint32_t *d; int *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 32); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, n);
    int r[8 * n]; hipMemcpy(r, o, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (r[8 * i] != r[8 * i + 1]) { if (bad++ < 4) printf("i %d ref %d got %d dl %d/%d dh %d/%d sc16 %08x d16 %08x scp %08x\n", i, r[8*i], r[8*i+1], r[8*i+2], r[8*i+3], r[8*i+4], r[8*i+5], r[8*i+6], r[8*i+7], h[4*i+3] & 0x3f3f3f3f); }
    printf("bad %d of %d\n", bad, n);
}
