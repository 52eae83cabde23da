// ubench_ceilings.hip -- the two measured ceilings bench.py reports beside the nominal peaks:
//   * fp64 vector issue rate per SIMD at 1 / 2 / 4 resident waves per SIMD (independent v_fma_f64 / v_add_f64 chains,
//     ILP 8): the blind-rotate kernels hold 2 waves per SIMD (256 VGPRs), and a SIMD with two waves does not reach
//     the 4-cycle cadence the nominal 78.6 Tflop/s assumes -- this is what "attainable at the kernel's occupancy" means;
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
