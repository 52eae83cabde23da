// ubench_mfma_f64.hip -- can the fp64 MATRIX pipe take a DFT stage of the negacyclic transform off the VALU?
//
// The blind-rotate kernels are bound by VALU issue slots + LDS stores while the matrix pipe idles.
// DESIGN 3.8 dismissed "a DFT level on v_mfma_f64_16x16x4_f64" on a flop count (a DFT-8 as a dense real 16 x 16 product costs
// 8.5 x the butterfly form and the fp64 matrix peak equals the vector peak on gfx950).  That argument ignores that the two
// pipes are separate: a wave issuing MFMAs beside a wave issuing butterflies might ADD throughput.  This measures it
// (poly/fourier_transform.go:178-347 is the transform in question).
//
// Unit of work: one "DFT-8 set" = 64 independent 8-point complex DFTs (what one wavefront holds at 8 points per lane).
//   VALU form: the register butterfly network on 8 complex fp64 per lane (3 radix-2 stages, the +-i and (1 +- i)/sqrt2 twiddles);
//              the compiled instruction count is printed by tools/run_ubench_mfma_f64.py from the ISA.
//   MFMA form: a DFT-8 on complex data is a real 16 x 16 matrix (re / im interleaved) times a real 16-vector; 16 columns = 16
//              DFT-8s per D[16x16] += A[16x4] B[4x16] chain of K = 16, i.e. FOUR v_mfma_f64_16x16x4_f64 per 16 DFT-8s and
//              SIXTEEN per DFT-8 set.  (Operand re-layout between the butterfly order and the MFMA fragments is NOT charged: this
//              is the upper bound of what the matrix pipe could contribute.)
// Configurations, all at 8 waves per CU = 2 per SIMD unless stated (the occupancy of the real kernels):
//   V2   two VALU waves per SIMD                      (the baseline: what the kernels do today)
//   V1   one VALU wave per SIMD
//   M1   one MFMA wave per SIMD,  M2  two MFMA waves per SIMD      (the matrix pipe's own rate)
//   VM   one VALU wave + one MFMA wave per SIMD       (the proposal)
//   VVM  two VALU waves + one MFMA wave per SIMD      (12 waves per CU: needs <= 168 VGPRs, which the real kernels do not have;
//                                                      shown as the additive upper bound)
// Output: DFT-8 sets per microsecond per CU for each configuration; the runner adds shader clock and socket power.
// Kill criterion: adopt only if VM >= 1.25 x V2 at the power-limited clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f64.hip -o tools/ubench_mfma_f64.bin ; run: <bin> <config> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double double4_ __attribute__((ext_vector_type(4)));
struct cd { double re, im; };

__device__ __forceinline__ void dft8(cd (&x)[8])
{
    // stage 1: (j, j+4)
    cd a[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a[j] = cd{x[j].re + x[j + 4].re, x[j].im + x[j + 4].im};
        a[j + 4] = cd{x[j].re - x[j + 4].re, x[j].im - x[j + 4].im};
    }
    // twiddles on the lower half: 1, w, -i, w^3 with w = (1 - i)/sqrt2
    const double r = 0.70710678118654752440;
    cd t5 = cd{(a[5].re + a[5].im) * r, (a[5].im - a[5].re) * r};
    cd t6 = cd{a[6].im, -a[6].re};
    cd t7 = cd{(a[7].im - a[7].re) * r, -(a[7].re + a[7].im) * r};
    a[5] = t5; a[6] = t6; a[7] = t7;
    // stage 2: (j, j+2) inside each half, twiddle -i on the odd pair
    cd b[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
        b[h + 0] = cd{a[h + 0].re + a[h + 2].re, a[h + 0].im + a[h + 2].im};
        b[h + 2] = cd{a[h + 0].re - a[h + 2].re, a[h + 0].im - a[h + 2].im};
        b[h + 1] = cd{a[h + 1].re + a[h + 3].re, a[h + 1].im + a[h + 3].im};
        cd d = cd{a[h + 1].re - a[h + 3].re, a[h + 1].im - a[h + 3].im};
        b[h + 3] = cd{d.im, -d.re};
    }
    // stage 3: (j, j+1)
#pragma unroll
    for (int h = 0; h < 8; h += 2) {
        x[h] = cd{b[h].re + b[h + 1].re, b[h].im + b[h + 1].im};
        x[h + 1] = cd{b[h].re - b[h + 1].re, b[h].im - b[h + 1].im};
    }
}

// role of a wave: 0 = VALU butterflies, 1 = MFMA chains, 2 = exit at once
template <int ROLES /* packed: 2 bits per wave group of 4 waves (one per SIMD) */, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(double *out, long iters, unsigned *simd_of_role)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int role = (ROLES >> (2 * (w >> 2))) & 3;
    if (lane == 0 && blockIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        simd_of_role[w] = ((hw >> 4) & 3) | (role << 8);
    }
    if (role == 0) {
        cd x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = cd{1.0 + lane + i, 0.5 * i - lane};
        for (long it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) dft8(x);                       // 4 DFT-8 sets per iteration
            if ((it & 15) == 15) {                                     // keep the values finite: |x| grows by sqrt8 per DFT
#pragma unroll
                for (int i = 0; i < 8; i++) { x[i].re *= 0x1p-96; x[i].im *= 0x1p-96; }
            }
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s += x[i].re + x[i].im;
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if (role == 1) {
        // four independent accumulator tiles (the chains of four different groups of 16 DFT-8s): MFMA back-to-back issue needs
        // independent destinations, as the butterflies of different points are independent
        double4_ c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
        for (long it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {                              // 4 x (4 tiles x K=4) = 64 MFMAs = 4 DFT-8 sets per iteration
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c3, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c3, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
            }
            if ((it & 63) == 63) { c0 *= 0x1p-60; c1 *= 0x1p-60; c2 *= 0x1p-60; c3 *= 0x1p-60; }
        }
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

struct Cfg { const char *name; int valu_waves, mfma_waves; };

template <int ROLES, int WAVES> static double run(const char *name, int nvalu, int nmfma, double seconds, int cus)
{
    double *out; unsigned *roles;
    hipMalloc(&out, (size_t)cus * WAVES * 64 * sizeof(double));
    hipMalloc(&roles, 64 * sizeof(unsigned)); hipMemset(roles, 0, 64 * sizeof(unsigned));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    long iters = 2000;
    float ms = 0;
    for (int pass = 0; pass < 2; pass++) {             // pass 0 calibrates the iteration count for the requested duration
        hipEventRecord(a);
        hipLaunchKernelGGL((k<ROLES, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, out, iters, roles);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        if (pass == 0) iters = (long)(iters * (seconds * 1e3 / ms)) + 1;
    }
    unsigned h[64]; hipMemcpy(h, roles, sizeof h, hipMemcpyDeviceToHost);
    int per_simd[4][3] = {};
    for (int w = 0; w < WAVES; w++) per_simd[h[w] & 3][(h[w] >> 8) & 3]++;
    const double sets = 4.0 * iters;                   // DFT-8 sets per wave
    const double v = nvalu * sets / (ms * 1e3), m = nmfma * sets / (ms * 1e3);
    printf("{\"config\": \"%s\", \"ms\": %.2f, \"valu_waves_per_cu\": %d, \"mfma_waves_per_cu\": %d, \"valu_sets_per_us_per_cu\": %.3f, "
           "\"mfma_sets_per_us_per_cu\": %.3f, \"total_sets_per_us_per_cu\": %.3f, \"mfma_tflops_chip\": %.1f, "
           "\"placement_valu_mfma_per_simd\": [[%d,%d],[%d,%d],[%d,%d],[%d,%d]]}\n",
           name, ms, nvalu, nmfma, v, m, v + m, m * 16 * 2048.0 * cus * 1e6 / 1e12,
           per_simd[0][0], per_simd[0][1], per_simd[1][0], per_simd[1][1], per_simd[2][0], per_simd[2][1], per_simd[3][0], per_simd[3][1]);
    fflush(stdout);
    hipFree(out); hipFree(roles);
    return v + m;
}

int main(int argc, char **argv)
{
    const char *cfg = argc > 1 ? argv[1] : "all";
    const double seconds = argc > 2 ? atof(argv[2]) : 0.5;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    auto want = [&](const char *n) { return !strcmp(cfg, "all") || !strcmp(cfg, n); };
    // ROLES: 2 bits per group of four waves (waves 4g .. 4g+3 land on SIMDs 0..3): 0 VALU, 1 MFMA, 2 idle
    if (want("V2")) run<0x0, 8>("V2", 8, 0, seconds, cus);
    if (want("V1")) run<0x0, 4>("V1", 4, 0, seconds, cus);
    if (want("M1")) run<0x1, 4>("M1", 0, 4, seconds, cus);
    if (want("M2")) run<0x5, 8>("M2", 0, 8, seconds, cus);
    if (want("VM")) run<0x4, 8>("VM", 4, 4, seconds, cus);
    if (want("VVM")) run<0x10, 12>("VVM", 8, 4, seconds, cus);
    return 0;
}
