// Micro-benchmarks backing DESIGN.md's cost model: fp64 VALU issue rate per SIMD and the
// wave-private LDS exchange round trip.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int ILP> __global__ void k_dp(double *out, int iters, double seed)
{
    double x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + i + threadIdx.x;
    const double a = 1.0000001, b = 0.9999999;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) x[i] = x[i] + a;
            else if (OP == 1) x[i] = x[i] * a;
            else if (OP == 2) x[i] = fma(x[i], a, b);
            else { float f = (float)x[i]; f = fmaf(f, 1.0000001f, 0.5f); x[i] = f; }
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

struct __attribute__((aligned(16))) cd { double re, im; };
__global__ void k_xchg(double *out, int iters)
{
    __shared__ cd sc[4][8 * 72];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 3, lo = lane & 7;
    cd x[8];
    for (int i = 0; i < 8; i++) x[i] = cd{(double)(lane + i), (double)i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 8; m++) sc[w][72 * m + lane] = x[m];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int b = 0; b < 8; b++) x[b] = sc[w][72 * hi + 8 * b + lo];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 8; i++) x[i].re += 1.0;     // dependent use
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i].re + x[i].im;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    double *out; hipMalloc(&out, sizeof(double) * 256 * 8 * 1024);
    const int iters = 20000;
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk_khz);
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "f32 fma (+cvt)"};
    for (int wps = 1; wps <= 4; wps *= 2) {                // waves per SIMD
        const int blocks = 256 * wps, threads = 256;      // 256 threads = 4 waves -> one per SIMD
#define RUN(OP, ILP) { float ms = timeit([&] { hipLaunchKernelGGL((k_dp<OP, ILP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0); }); \
        double inst = (double)iters * ILP * wps; printf("waves/SIMD %d  %-14s ILP %d : %.3f ms  -> %.2f cycles/instr/SIMD @2.4GHz\n", wps, names[OP], ILP, ms, ms * 1e-3 * 2.4e9 / inst); }
        RUN(0, 8) RUN(1, 8) RUN(2, 8) RUN(0, 2) RUN(2, 16)
    }
    for (int wpc = 4; wpc <= 16; wpc *= 2) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_xchg, dim3(256 * wpc / 4), dim3(256), 0, 0, out, 5000); });
        printf("LDS exchange round trip, %2d waves/CU: %.1f cycles per exchange per wave (wall/iters @2.4GHz)\n", wpc, ms * 1e-3 * 2.4e9 / 5000);
    }
    return 0;
}
