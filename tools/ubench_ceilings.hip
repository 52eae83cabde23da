// ubench_ceilings.hip -- the two measured ceilings bench.py reports beside the nominal peaks:
//   * fp64 vector issue rate per SIMD at 1 / 2 / 4 resident waves per SIMD (independent v_fma_f64 / v_add_f64 chains,
//     ILP 8): the blind-rotate kernels hold 2 waves per SIMD (256 VGPRs), and a SIMD with two waves does not reach
//     the 4-cycle cadence the nominal 78.6 Tflop/s assumes -- this is what "attainable at the kernel's occupancy" means;
//   * LDS instruction throughput per CU for the four DS instruction kinds of the blind-rotate kernels (the store path is the
//     narrow one: a ds_write_b128 costs ~14 cycles of CU time, a ds_read_b128 ~4);
//   * device-to-device copy bandwidth (a plain 16 B/lane copy kernel and hipMemcpyDtoD over 1 GiB): the HBM rate this
//     box actually delivers, next to the nominal 8 TB/s.
// Prints ONE JSON object.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ceilings.hip -o tools/ubench_ceilings.bin
// (done by __graft_entry__.build()); run by bench.py after its timed region, or by hand on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP, int ILP> __global__ void k_dp(double *out, int iters, double seed)
{
    double x[ILP];
    for (int i = 0; i < ILP; i++) x[i] = seed + i + threadIdx.x;
    const double a = 1.0000001, b = 0.9999999;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) x[i] = OP == 0 ? x[i] + a : fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += x[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS instruction throughput per CU (tools/ubench_ldsrate.hip has the full table): bursts of 32 conflict-free DS instructions of
// one kind per wave, 8 waves per CU.  KIND 0 ds_read_b32, 1 ds_read_b128, 2 ds_write_b128, 3 ds_add_u32.
template <int KIND> __global__ __launch_bounds__(256) void k_lds(uint32_t *out, int iters)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint4 buf[4096];                       // 64 KB: two workgroups per CU
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)&buf[w * 1024];
    const uint32_t a32 = base + lane * 4, a128 = base + lane * 16;
    uint32_t r0 = 0, acc = 0;
    u32x4 r2 = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 32; c++) {
            if (KIND == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0) : "v"(a32), "n"((c & 15) * 256));
            if (KIND == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2) : "v"(a128), "n"((c & 15) * 1024));
            if (KIND == 2) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a128), "v"(r2), "n"((c & 15) * 1024));
            if (KIND == 3) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(a32), "v"(acc), "n"((c & 15) * 256));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += r0 + r2.x;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <class F> static float best_ms(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    double *out; CHECK(hipMalloc(&out, sizeof(double) * (size_t)cus * 16 * 256));
    const int iters = 20000, ILP = 8;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"fp64_issue\": {", prop.gcnArchName, cus, prop.clockRate);
    bool first = true;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * wps;                       // 256 threads = 4 waves = one per SIMD
        for (int op = 0; op < 2; op++) {
            float ms = op == 0 ? best_ms([&] { hipLaunchKernelGGL((k_dp<0, ILP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, 3)
                               : best_ms([&] { hipLaunchKernelGGL((k_dp<1, ILP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, 3);
            // wave-instructions per SIMD = iters * ILP * wps; a wave64 instruction = 64 lanes
            const double inst_per_simd = (double)iters * ILP * wps, ns_per_inst = ms * 1e6 / inst_per_simd;
            const double lane_ops_per_s = inst_per_simd * simds * 64.0 / (ms * 1e-3);
            printf("%s\"%s_wps%d\": {\"ns_per_instr_per_simd\": %.4f, \"T_lane_instr_per_s\": %.3f, \"Tflops\": %.2f}", first ? "" : ", ",
                   op == 0 ? "add" : "fma", wps, ns_per_inst, lane_ops_per_s / 1e12, lane_ops_per_s * (op == 0 ? 1 : 2) / 1e12);
            first = false;
        }
    }
    printf("}, \"lds_ns_per_wave_instr_per_cu\": {");
    {
        uint32_t *lout; CHECK(hipMalloc(&lout, (size_t)cus * 2 * 256 * 4));
        const int it2 = 2000;
        const double ops_per_cu = 2.0 * 4 * it2 * 32;          // two workgroups of four waves per CU
        float t0 = best_ms([&] { hipLaunchKernelGGL(k_lds<0>, dim3(cus * 2), dim3(256), 0, 0, lout, it2); }, 3);
        float t1 = best_ms([&] { hipLaunchKernelGGL(k_lds<1>, dim3(cus * 2), dim3(256), 0, 0, lout, it2); }, 3);
        float t2 = best_ms([&] { hipLaunchKernelGGL(k_lds<2>, dim3(cus * 2), dim3(256), 0, 0, lout, it2); }, 3);
        float t3 = best_ms([&] { hipLaunchKernelGGL(k_lds<3>, dim3(cus * 2), dim3(256), 0, 0, lout, it2); }, 3);
        printf("\"ds_read_b32\": %.4f, \"ds_read_b128\": %.4f, \"ds_write_b128\": %.4f, \"ds_add_u32\": %.4f", t0 * 1e6 / ops_per_cu,
               t1 * 1e6 / ops_per_cu, t2 * 1e6 / ops_per_cu, t3 * 1e6 / ops_per_cu);
        (void)hipFree(lout);
    }
    printf("}, ");
    const size_t bytes = (size_t)1 << 30;
    void *src, *dst;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 2, bytes));
    float k_ms = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(cus * 16), dim3(256), 0, 0, (const uint4 *)src, (uint4 *)dst, bytes / 16); }, 5);
    float m_ms = best_ms([&] { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
    // a copy reads and writes every byte: traffic = 2 x bytes
    printf("\"hbm_copy\": {\"bytes\": %zu, \"kernel_copy_GBps\": %.1f, \"memcpy_d2d_GBps\": %.1f, \"note\": \"read + write traffic, 2 x bytes / time\"}}\n",
           bytes, 2.0 * bytes / (k_ms * 1e-3) / 1e9, 2.0 * bytes / (m_ms * 1e-3) / 1e9);
    return 0;
}
