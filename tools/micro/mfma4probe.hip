// Lane layout probe for v_mfma_i32_4x4x4_16b_i8 on gfx950: prints, for every lane and result register, which
// (block, row, col) of D = A x B it holds, assuming operand lane l supplies A[block(l)][row(l)][k = 0..3] as the bytes of
// one dword and B[block(l)][k][col(l)] likewise.  build: hipcc --offload-arch=gfx950 -O2 mfma4probe.hip -o mfma4probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(int *o, const int *a, const int *b) {
    i32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_4x4x4i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) o[threadIdx.x * 4 + r] = c[r];
}
int main() {
    int ha[64], hb[64], ho[256];
    // A lane l: bytes k -> value (l + 1) * (k == l % 4 ... ) : use unique small primes so that products identify (la, lb)
    // simpler: A lane l = all four bytes equal to (l % 8 + 1) with a one-hot k pattern is ambiguous; use two passes.
    int *da, *db, *dout;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024);
    // pass 1: A lane l: byte k = (k == 0 ? l + 1 : 0);  B lane l: byte k = (k == 0 ? 1 : 0)  -> D = sum over the A lanes that feed it
    // pass 2: swapped roles.  From pass 1, D[lane][reg] = (la + 1) identifies the A lane; from pass 2 the B lane.
    for (int pass = 0; pass < 2; pass++) {
        for (int l = 0; l < 64; l++) { ha[l] = pass == 0 ? (l + 1) : 1; hb[l] = pass == 0 ? 1 : (l + 1); }
        hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, da, db);
        hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
        printf("pass %d (%s lane feeding each result):\n", pass, pass == 0 ? "A" : "B");
        for (int l = 0; l < 64; l++) printf("  lane %2d: %3d %3d %3d %3d\n", l, ho[l * 4] - 1, ho[l * 4 + 1] - 1, ho[l * 4 + 2] - 1, ho[l * 4 + 3] - 1);
    }
    // pass 3: k pairing: A byte k = 1 << (2k) in lane 0's block row..., B byte k = k + 1: D = sum_k A_k * B_k = sum (k+1) << 2k = 1 + 8 + 48 + 256 = 313 if byte k meets byte k
    for (int l = 0; l < 64; l++) { ha[l] = 0x40100401; hb[l] = 0x04030201; } // bytes (k): A = {1, 4, 16, 64}, B = {1, 2, 3, 4}
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, da, db);
    hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    printf("pass 3: lane 0 reg 0 = %d (321 = byte k of A meets byte k of B: 1*1 + 4*2 + 16*3 + 64*4)\n", ho[0]);
    return 0;
}
