// Lane layout of v_mfma_i32_32x32x32_i8 on gfx950, checked against a host product:
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_i8.hip -o /tmp/probe_mfma_i8 && /tmp/probe_mfma_i8
// Hypothesis (as the f16 32x32x16 form, K doubled): lane l holds A[row l&31][k = 16 (l>>5) .. +15] and
// B[k = 16 (l>>5) .. +15][col l&31] as 16 packed bytes; D[row (r&3) + 8 (r>>2) + 4 (l>>5)][col l&31] in register r.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void probe(const int8_t *A /*[32][32]*/, const int8_t *B /*[32][32] k-major*/, int *D /*[32][32]*/)
{
    const int l = threadIdx.x;
    v4i a, b;
    int8_t ab[16], bb[16];
    for (int x = 0; x < 16; x++) { ab[x] = A[(l & 31) * 32 + 16 * (l >> 5) + x]; bb[x] = B[(16 * (l >> 5) + x) * 32 + (l & 31)]; }
    __builtin_memcpy(&a, ab, 16); __builtin_memcpy(&b, bb, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main()
{
    int8_t hA[1024], hB[1024]; int hD[1024], ref[1024];
    srand(7);
    for (int i = 0; i < 1024; i++) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { int s = 0; for (int k = 0; k < 32; k++) s += (int)hA[m * 32 + k] * (int)hB[k * 32 + n]; ref[m * 32 + n] = s; }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += hD[i] != ref[i];
    printf("mismatches: %d of 1024 (signed x signed, hypothesis %s)\n", bad, bad ? "WRONG" : "confirmed");
    return bad != 0;
}
