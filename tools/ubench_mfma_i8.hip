// Issue rate of v_mfma_i32_32x32x32_i8 on gfx950: 16 independent accumulator tiles per wave, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_i8.hip -o /tmp/ubench_mfma_i8 && /tmp/ubench_mfma_i8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void loop(int iters, int *sink)
{
    v16i acc[NACC];
    for (int q = 0; q < NACC; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[q], 0, 0, 0);
    }
    int s = 0;
    for (int q = 0; q < NACC; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
    if (s == 0x12345678) sink[0] = s;
}
template <int NACC> void run(int wgs, const char *what)
{
    int *sink; (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4096;
    hipLaunchKernelGGL(loop<NACC>, dim3(wgs), dim3(256), 0, 0, 16, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(loop<NACC>, dim3(wgs), dim3(256), 0, 0, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * NACC, ops = mfma_per_wave * 4.0 * wgs * 65536.0;
    printf("%-34s %7.3f ms  %6.1f ns per MFMA per wave  %7.1f TOPS\n", what, ms, ms * 1e6 / mfma_per_wave, ops / (ms * 1e-3) / 1e12);
}
int main()
{
    run<16>(256, "16 acc tiles, 1 WG(4 waves)/CU");
    run<4>(256, "4 acc tiles, 1 WG/CU");
    run<4>(512, "4 acc tiles, 2 WG/CU");
    run<4>(1024, "4 acc tiles, 4 WG/CU");
    return 0;
}
