// Does LDS exchange traffic from one wave slow the fp64 VALU stream of another wave on the same SIMD / CU?
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((aligned(16))) cd { double re, im; };
#define ORDER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// mode: 0 = all waves fp64, 1 = all waves LDS exchange, 2 = even waves fp64 / odd waves LDS, 3 = each wave alternates (fp64 block, exchange)
template <int MODE, int WIDE> __global__ __launch_bounds__(512) void k(double *out, int iters, long long *cyc)
{
    __shared__ cd sc[8][8 * 72];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 3, lo = lane & 7;
    cd x[8]; double y[8];
    for (int i = 0; i < 8; i++) { x[i] = cd{(double)(lane + i), (double)i}; y[i] = lane + i; }
    const bool do_dp = MODE == 0 || MODE == 3 || (MODE == 2 && (w & 1) == 0);
    const bool do_lds = MODE == 1 || MODE == 3 || (MODE == 2 && (w & 1) == 1);
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (do_dp) {
#pragma unroll
            for (int r = 0; r < 12; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = fma(y[i], 1.0000001, 0.5);
        }
        if (do_lds) {
            if (WIDE) {
#pragma unroll
                for (int m = 0; m < 8; m++) sc[w][72 * m + lane] = x[m];
                ORDER();
#pragma unroll
                for (int b = 0; b < 8; b++) x[b] = sc[w][72 * hi + 8 * b + lo];
                ORDER();
            } else {   // same bytes as ds_write_b64 / ds_read_b64 pairs
                double *s = reinterpret_cast<double *>(&sc[w][0]);
#pragma unroll
                for (int m = 0; m < 8; m++) { s[(72 * m + lane)] = x[m].re; s[576 + 72 * m + lane] = x[m].im; }
                ORDER();
#pragma unroll
                for (int b = 0; b < 8; b++) { x[b].re = s[72 * hi + 8 * b + lo]; x[b].im = s[576 + 72 * hi + 8 * b + lo]; }
                ORDER();
            }
#pragma unroll
            for (int i = 0; i < 8; i++) x[i].re += 1.0;
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i].re + x[i].im + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 64 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int MODE, int WIDE> void run(const char *name, double *out, long long *cyc)
{
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE, WIDE>), dim3(256), dim3(512), 0, 0, out, iters, cyc); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL((k<MODE, WIDE>), dim3(256), dim3(512), 0, 0, out, iters, cyc); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s %.3f ms   wave0 %.0f clk/iter  wave1 %.0f clk/iter\n", name, ms, (double)h[0] / iters, (double)h[1] / iters);
}
int main()
{
    double *out; long long *cyc; hipMalloc(&out, 8 * 256 * 512); hipMalloc(&cyc, 16);
    printf("8 waves/CU (2 per SIMD); per iter: fp64 block = 96 v_fma_f64, exchange = 8 KiB written + 8 KiB read\n");
    run<0, 1>("all waves fp64", out, cyc);
    run<1, 1>("all waves exchange (b128)", out, cyc);
    run<1, 0>("all waves exchange (b64)", out, cyc);
    run<2, 1>("even waves fp64, odd waves exchange (b128)", out, cyc);
    run<3, 1>("every wave: fp64 block then exchange (b128)", out, cyc);
    run<3, 0>("every wave: fp64 block then exchange (b64)", out, cyc);
    return 0;
}
