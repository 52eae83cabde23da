// ubench_ldsbank.hip -- how many LDS banks a ds_read_b32 sees on gfx950, and whether equal addresses broadcast:
// every lane reads word (row[lane] * 65 + c) for c = 0..63, with row[] = identity / a permutation / uniformly random over 64
// or 32 rows / one row for all lanes; compared with the lane-contiguous pattern.  Prints ns per wave-read-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsbank.hip -o /tmp/ubench_ldsbank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k(const int *rows, uint32_t *out, int iters, int stride, int contiguous)
{
    __shared__ uint32_t buf[64 * 66];
    for (int i = threadIdx.x; i < 64 * 66; i += 256) buf[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int r = rows[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane];
    uint32_t acc = 0;
    const uint32_t base = (uint32_t)(size_t)&buf[contiguous ? lane : r * stride];
    const uint32_t step = contiguous ? 256 : 4;
    for (int it = 0; it < iters; it++) {
        uint32_t v[16];
#pragma unroll
        for (int c = 0; c < 16; c++) asm volatile("ds_read_b32 %0, %1" : "=v"(v[c]) : "v"(base + step * (c + 16 * (it & 3))));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 16; c++) acc += v[c];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int blocks = 256 * 4, iters = 4000;
    std::vector<int> h(blocks * 4 * 64);
    int *d_rows; uint32_t *d_out;
    hipMalloc(&d_rows, h.size() * 4); hipMalloc(&d_out, blocks * 256 * 4);
    auto run = [&](const char *name, int stride, int contiguous) {
        hipMemcpy(d_rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_rows, d_out, iters, stride, contiguous);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_rows, d_out, iters, stride, contiguous);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per CU: blocks/256 workgroups x 4 waves x iters x 16 reads
        const double reads_per_cu = (double)blocks / 256 * 4 * iters * 16;
        printf("%-44s stride %2d: %.3f ms  -> %.2f ns per wave-read per CU (%.1f cycles @2.4GHz)\n", name, stride, ms, ms * 1e6 / reads_per_cu,
               ms * 1e6 / reads_per_cu * 2.4);
    };
    for (size_t i = 0; i < h.size(); i++) h[i] = i & 63;
    run("lane-contiguous (reference)", 65, 1);
    run("row = lane (all distinct)", 65, 0);
    run("row = lane (all distinct)", 64, 0);
    for (size_t i = 0; i < h.size(); i++) h[i] = 5;
    run("one row for all lanes (broadcast)", 65, 0);
    srand(1);
    for (size_t i = 0; i < h.size(); i++) h[i] = rand() & 63;
    run("random rows of 64", 65, 0);
    run("random rows of 64", 66, 0);
    for (size_t i = 0; i < h.size(); i++) h[i] = rand() & 31;
    run("random rows of 32", 65, 0);
    for (size_t i = 0; i < h.size(); i++) h[i] = (i & 31);
    run("row = lane mod 32 (pairs share a row)", 65, 0);
    for (size_t i = 0; i < h.size(); i++) h[i] = (i & 63) ^ 32;
    run("row = lane ^ 32", 65, 0);
    return 0;
}
