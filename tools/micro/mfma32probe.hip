// Layout check for v_mfma_i32_16x16x32_i8 on gfx950 (the operand layout the Q4_K batched mat-mul relies on):
//   A operand of lane l = the 8 bytes A[i = l % 16][k = 8 * (l / 16) .. + 7]      (byte b of the 64-bit operand = k offset b)
//   B operand of lane l = the 8 bytes B[k = 8 * (l / 16) .. + 7][j = l % 16]
//   D register r of lane l = D[i = 4 * (l / 16) + r][j = l % 16]
// Random asymmetric signed bytes; prints the number of mismatches against the CPU product (0 = layout confirmed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(int *o, const long *a, const long *b) {
    i32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) o[threadIdx.x * 4 + r] = c[r];
}
int main() {
    int8_t A[16][32], B[32][16];
    srand(7);
    for (int i = 0; i < 16; i++) for (int kk = 0; kk < 32; kk++) A[i][kk] = (int8_t)(rand() % 255 - 127);
    for (int kk = 0; kk < 32; kk++) for (int j = 0; j < 16; j++) B[kk][j] = (int8_t)(rand() % 255 - 127);
    long ha[64], hb[64];
    for (int l = 0; l < 64; l++) {
        uint64_t va = 0, vb = 0;
        for (int b = 0; b < 8; b++) {
            va |= (uint64_t)(uint8_t)A[l % 16][8 * (l / 16) + b] << (8 * b);
            vb |= (uint64_t)(uint8_t)B[8 * (l / 16) + b][l % 16] << (8 * b);
        }
        ha[l] = (long)va; hb[l] = (long)vb;
    }
    long *da, *db; int *dout; int ho[256];
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dout, 1024);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, da, db);
    hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        const int i = 4 * (l / 16) + r, j = l % 16;
        int ref = 0;
        for (int kk = 0; kk < 32; kk++) ref += (int)A[i][kk] * (int)B[kk][j];
        if (ref != ho[l * 4 + r]) bad++;
    }
    printf("v_mfma_i32_16x16x32_i8 layout check: %d mismatches of 256\n", bad);
    return 0;
}
