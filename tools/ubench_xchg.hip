// Register-path (permlane swap + DPP) vs LDS-path lane<->register exchange of 8 complex fp64 per lane,
// alone and mixed with an fp64 block (8 waves/CU = 2 per SIMD).  Also checks the two paths agree.
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((aligned(16))) cd { double re, im; };
#define ORDER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
union U { cd c; unsigned w[4]; };

__device__ __forceinline__ void swap32(cd &a, cd &b) { U x, y; x.c = a; y.c = b;
#pragma unroll
    for (int i = 0; i < 4; i++) { auto r = __builtin_amdgcn_permlane32_swap(x.w[i], y.w[i], false, false); x.w[i] = r[0]; y.w[i] = r[1]; }
    a = x.c; b = y.c; }
__device__ __forceinline__ void swap16(cd &a, cd &b) { U x, y; x.c = a; y.c = b;
#pragma unroll
    for (int i = 0; i < 4; i++) { auto r = __builtin_amdgcn_permlane16_swap(x.w[i], y.w[i], false, false); x.w[i] = r[0]; y.w[i] = r[1]; }
    a = x.c; b = y.c; }
template <int SH, int MHI, int MLO> __device__ __forceinline__ void swapdpp(cd &a, cd &b) { U x, y, nx, ny; x.c = a; y.c = b;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        nx.w[i] = __builtin_amdgcn_update_dpp(x.w[i], y.w[i], 0x110 + SH, 0xF, MHI, false);   // row_shr: lanes with the bit set take b[l-SH]
        ny.w[i] = __builtin_amdgcn_update_dpp(y.w[i], x.w[i], 0x100 + SH, 0xF, MLO, false);   // row_shl: lanes with the bit clear take a[l+SH]
    }
    a = nx.c; b = ny.c; }
// exchange 1: reg bits [2:0] <-> lane bits [5:3]
__device__ __forceinline__ void xchg1_reg(cd (&x)[8]) {
#pragma unroll
    for (int r = 0; r < 4; r++) swap32(x[r], x[r + 4]);
    swap16(x[0], x[2]); swap16(x[1], x[3]); swap16(x[4], x[6]); swap16(x[5], x[7]);
    swapdpp<8, 0xC, 0x3>(x[0], x[1]); swapdpp<8, 0xC, 0x3>(x[2], x[3]); swapdpp<8, 0xC, 0x3>(x[4], x[5]); swapdpp<8, 0xC, 0x3>(x[6], x[7]);
}
__device__ __forceinline__ void xchg1_lds(cd (&x)[8], cd *sc, int lane) {
    const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
    for (int m = 0; m < 8; m++) sc[72 * m + lane] = x[m];
    ORDER();
#pragma unroll
    for (int b = 0; b < 8; b++) x[b] = sc[72 * hi + 8 * b + lo];
    ORDER();
}
template <int MODE> __global__ __launch_bounds__(512) void k(double *out, int iters, int *bad)
{
    __shared__ cd sc[8][8 * 72];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    cd x[8]; double y[8];
    for (int i = 0; i < 8; i++) { x[i] = cd{(double)(lane * 8 + i), (double)(1000 + lane * 8 + i)}; y[i] = lane + i; }
    if (MODE == 9) {            // correctness: both paths must give the same permutation
        cd a[8], b[8];
        for (int i = 0; i < 8; i++) a[i] = b[i] = x[i];
        xchg1_reg(a); xchg1_lds(b, sc[w], lane);
        for (int i = 0; i < 8; i++) if (a[i].re != b[i].re || a[i].im != b[i].im) atomicAdd(bad, 1);
        return;
    }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 12; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = fma(y[i], 1.0000001, 0.5);
        }
        if (MODE == 1 || MODE == 3) xchg1_reg(x);
        if (MODE == 4 || MODE == 2) xchg1_lds(x, sc[w], lane);
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) x[i].re += 1.0;
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i].re + x[i].im + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, double *out, int *bad)
{
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, iters, bad); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, iters, bad); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-48s %.3f ms  (%.0f cycles/iter/SIMD-pair @2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}
int main()
{
    double *out; int *bad; hipMalloc(&out, 8 * 256 * 512); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k<9>), dim3(4), dim3(512), 0, 0, out, 1, bad); hipDeviceSynchronize();
    int h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("register exchange vs LDS exchange mismatches: %d\n", h);
    run<0>("fp64 block only (96 fma)", out, bad);
    run<1>("register exchange only (64 VALU)", out, bad);
    run<4>("LDS exchange only", out, bad);
    run<3>("fp64 block + register exchange", out, bad);
    run<2>("fp64 block + LDS exchange", out, bad);
    return 0;
}
