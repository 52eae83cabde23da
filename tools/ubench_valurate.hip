// ubench_valurate.hip -- issue cost of the lane-crossing VALU instructions a register-path FFT exchange would be made of
// (v_permlane32_swap, v_permlane16_swap, v_mov_b32_dpp, v_cndmask_b32_dpp) next to v_mov_b32 / v_add_f64 / v_fma_f64,
// as cycles of SIMD time per wave-instruction at 1, 2 and 4 waves per SIMD.  Streams of 32 instructions on 16 independent
// register pairs, no memory.     Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valurate.hip -o tools/ubench_valurate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND> __global__ __launch_bounds__(1024) void k(uint32_t *out, int iters)
{
    uint32_t a[16], b[16];
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 16 + i; b[i] = threadIdx.x * 7 + i; d[i] = (double)a[i]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 2; rep++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                if (KIND == 1) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
                if (KIND == 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
                if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 4) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]));
                if (KIND == 5) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]), "v"(a[(i + 1) & 15]) : "vcc");
                if (KIND == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
                if (KIND == 7) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
                if (KIND == 8) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(a[(i + 1) & 15]) : "vcc");
                if (KIND == 9) asm volatile("v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(a[i]) : "v"(b[i]));
            }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + b[i] + (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> static void run(const char *name, uint32_t *out, int cus, double ghz)
{
    printf("%-52s", name);
    for (int waves = 4; waves <= 16; waves *= 2) {         // per CU = 1, 2, 4 per SIMD
        const int iters = 4000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(64 * waves), 0, 0, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(64 * waves), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double per_simd = (double)(waves / 4) * iters * 32;
        printf("  %d/SIMD: %5.2f", waves / 4, ms * 1e-3 * ghz * 1e9 / per_simd);
    }
    printf("   cycles per wave-instruction per SIMD\n");
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    uint32_t *out; hipMalloc(&out, (size_t)p.multiProcessorCount * 1024 * 4);
    printf("%s, %d CUs, %.2f GHz (nominal clock used for the cycle figures)\n", p.gcnArchName, p.multiProcessorCount, ghz);
    run<0>("v_mov_b32", out, p.multiProcessorCount, ghz);
    run<8>("v_cndmask_b32", out, p.multiProcessorCount, ghz);
    run<1>("v_permlane32_swap_b32", out, p.multiProcessorCount, ghz);
    run<2>("v_permlane16_swap_b32", out, p.multiProcessorCount, ghz);
    run<3>("v_mov_b32_dpp row_ror:8 bank_mask:0x3", out, p.multiProcessorCount, ghz);
    run<9>("v_mov_b32_dpp row_shl:4 bank_mask:0x5", out, p.multiProcessorCount, ghz);
    run<4>("v_mov_b32_dpp quad_perm:[1,0,3,2]", out, p.multiProcessorCount, ghz);
    run<5>("v_cndmask_b32_dpp row_ror:8", out, p.multiProcessorCount, ghz);
    run<6>("v_add_f64", out, p.multiProcessorCount, ghz);
    run<7>("v_fma_f64", out, p.multiProcessorCount, ghz);
    return 0;
}
