// ubench_ldsrate.hip -- LDS instruction THROUGHPUT per CU on gfx950 (not latency): every wave issues bursts of 32 conflict-free
// DS instructions of one kind (lane-contiguous addresses) and waits once per burst; 4 / 8 / 16 waves per CU.
// Prints cycles of CU time per wave-instruction and the implied bytes per clock.  This is the number the blind-rotate
// kernels are bound by.   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsrate.hip -o /tmp/u
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND> __global__ __launch_bounds__(256) void k(uint32_t *out, int iters)
{
    __shared__ uint4 buf[4096];                       // 64 KB
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    uint32_t base = (uint32_t)(size_t)&buf[w * 1024];
    uint32_t a32 = base + lane * 4, a64 = base + lane * 8, a128 = base + lane * 16;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    uint32_t r0 = 0; uint64_t r1 = 0, r1b = 1; u32x4 r2 = {0, 0, 0, 0};
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 32; c++) {
            if (KIND == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0) : "v"(a32), "n"((c & 15) * 256));
            if (KIND == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r1) : "v"(a64), "n"((c & 15) * 512));
            if (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2) : "v"(a128), "n"((c & 15) * 1024));
            if (KIND == 3) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a32), "v"(acc), "n"((c & 15) * 256));
            if (KIND == 4) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a64), "v"(r1), "n"((c & 15) * 512));
            if (KIND == 5) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a128), "v"(r2), "n"((c & 15) * 1024));
            if (KIND == 6) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(a32), "v"(acc), "n"((c & 15) * 256));
            if (KIND == 7) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a64), "v"(r1), "v"(r1b), "n"(c & 7), "n"(16 + (c & 7)));
            if (KIND == 8) asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r2) : "v"(a64), "n"(c & 7), "n"(16 + (c & 7)));
            if (KIND == 9) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a128), "v"(r1), "v"(r1b), "n"(0), "n"(1));
            if (KIND == 10) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r2) : "v"(a128), "n"(0), "n"(1));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += r0 + (uint32_t)r1 + r2.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND> static void run(const char *name, int bytes, uint32_t *out, int cus)
{
    for (int wgs = 1; wgs <= 2; wgs++) {               // workgroups (of 4 waves, 64 KB LDS) per CU
        const int iters = 2000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wgs), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(cus * wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ops_per_cu = (double)wgs * 4 * iters * 32;
        const double cyc = ms * 1e-3 * 2.4e9 / ops_per_cu;
        printf("%-44s %2d waves/CU: %6.2f cycles per wave-instruction per CU  (%5.1f B/clk)\n", name, 4 * wgs, cyc, bytes * 64 / cyc);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    uint32_t *out; hipMalloc(&out, (size_t)p.multiProcessorCount * 2 * 256 * 4);
    run<0>("ds_read_b32", 4, out, p.multiProcessorCount);
    run<1>("ds_read_b64", 8, out, p.multiProcessorCount);
    run<2>("ds_read_b128", 16, out, p.multiProcessorCount);
    run<3>("ds_write_b32", 4, out, p.multiProcessorCount);
    run<4>("ds_write_b64", 8, out, p.multiProcessorCount);
    run<5>("ds_write_b128", 16, out, p.multiProcessorCount);
    run<6>("ds_add_u32", 4, out, p.multiProcessorCount);
    run<7>("ds_write2st64_b64 (16 B/lane, two planes)", 16, out, p.multiProcessorCount);
    run<8>("ds_read2st64_b64 (16 B/lane, two planes)", 16, out, p.multiProcessorCount);
    run<9>("ds_write2_b64 (16 B/lane, adjacent)", 16, out, p.multiProcessorCount);
    run<10>("ds_read2_b64 (16 B/lane, adjacent)", 16, out, p.multiProcessorCount);
    return 0;
}
