// ubench_bpermute.hip -- does ds_bpermute_b32 (and ds_permute_b32 / ds_swizzle_b32) ride the LDS LOAD path (~244 B/clk per CU on
// gfx950, tools/ubench_ldsrate.hip) or the STORE path (~73 B/clk)?  It writes no LDS memory, and the store path is the other
// binding resource of the blind-rotate kernels, so the answer prices a register-path FFT exchange
// (poly/fourier_transform.go:178-347 is what these exchanges implement) built from bpermutes.
//
// Part 1: throughput per CU of one DS kind at a time, bursts of 32, at 8 and 16 waves per CU (same harness as ubench_ldsrate,
//         so the read_b32 / write_b32 / read_b128 / write_b128 lines calibrate it).
// Part 2: exchange 1 of the radix-8 transform -- (reg m; lane 8b + c) <-> (reg b; lane 8m + c) on 8 complex fp64 per lane --
//         built as "rotate registers by the lane's row (3 stages of selects), 32 bpermutes, rotate back", checked against the
//         LDS form (8 ds_write_b128 + 8 ds_read_b128), timed alone and under an fp64 block at two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_bpermute.hip -o /tmp/u
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND> __global__ __launch_bounds__(256) void k_rate(uint32_t *out, int iters)
{
    __shared__ uint4 buf[4096];                       // 64 KB
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    uint32_t base = (uint32_t)(size_t)&buf[w * 1024];
    uint32_t a32 = base + lane * 4, a128 = base + lane * 16;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    uint32_t r0 = 0; u32x4 r2 = {0, 0, 0, 0};
    uint32_t acc = lane, src = (uint32_t)(((lane * 9) & 63) * 4);          // bpermute address: byte offset of the source lane
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 32; c++) {
            if (KIND == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0) : "v"(a32), "n"((c & 15) * 256));
            if (KIND == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a32), "v"(acc), "n"((c & 15) * 256));
            if (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2) : "v"(a128), "n"((c & 15) * 1024));
            if (KIND == 3) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a128), "v"(r2), "n"((c & 15) * 1024));
            if (KIND == 4) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r0) : "v"(src), "v"(acc));
            if (KIND == 5) asm volatile("ds_permute_b32 %0, %1, %2" : "=v"(r0) : "v"(src), "v"(acc));
            if (KIND == 6) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,8)" : "=v"(r0) : "v"(acc));
            if (KIND == 7) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r0) : "v"((uint32_t)(lane * 4)), "v"(acc));      // identity: no lane conflicts at all
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += r0 + r2.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND> static void rate(const char *name, int bytes, uint32_t *out, int cus)
{
    for (int wgs = 2; wgs <= 4; wgs += 2) {            // workgroups of 4 waves per CU: 8 and 16 waves per CU
        const int iters = 2000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(cus * wgs), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(cus * wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ops_per_cu = (double)wgs * 4 * iters * 32;
        const double cyc = ms * 1e-3 * 2.4e9 / ops_per_cu;
        printf("%-44s %2d waves/CU: %6.2f cycles per wave-instruction per CU  (%5.1f B/clk)\n", name, 4 * wgs, cyc, bytes * 64 / cyc);
    }
}

// ---- part 2: the exchange
struct __attribute__((aligned(16))) cd { double re, im; };
union U { cd c; unsigned w[4]; };
#define ORDER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ void xchg1_lds(cd (&x)[8], cd *sc, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
    for (int m = 0; m < 8; m++) sc[72 * m + lane] = x[m];
    ORDER();
#pragma unroll
    for (int b = 0; b < 8; b++) x[b] = sc[72 * hi + 8 * b + lo];
    ORDER();
}

// want: new x[b] at lane (hi, lo) = old x[hi] at lane (b, lo).
// 1. rotate left by hi on the SOURCE side: y[j] = x[(j + hi) & 7]               (3 stages x 8 regs x 4 dwords of v_cndmask)
//    -> lane (s, lo) holds in y[j] its x[(j + s) & 7]
// 2. z[j] at lane (hi, lo) = y[j] of lane ((hi - j) & 7, lo) = x[hi] of that lane   (one bpermute per register dword: 32)
// 3. new x[b] = z[(hi - b) & 7]: rotate by -hi with index reversal                  (3 stages of selects + a static reversal)
template <int S> __device__ __forceinline__ void rot_stage(cd (&x)[8], bool take)      // x[j] <- take ? x[(j + S) & 7] : x[j]
{
    cd y[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const double are = x[(j + S) & 7].re, aim = x[(j + S) & 7].im, bre = x[j].re, bim = x[j].im;   // values first: a select of
        y[j].re = take ? are : bre;                                                                    // ADDRESSES would send the
        y[j].im = take ? aim : bim;                                                                    // array to scratch
    }
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = y[j];
}
__device__ __forceinline__ void rot_by(cd (&x)[8], int amount)                     // x[j] <- x[(j + amount) & 7], amount per lane
{
    rot_stage<1>(x, amount & 1);
    rot_stage<2>(x, amount & 2);
    rot_stage<4>(x, amount & 4);
}
__device__ __forceinline__ void xchg1_bperm(cd (&x)[8], int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    rot_by(x, hi);
    cd z[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int srcl = (((hi - j) & 7) << 3) | lo;
        const int a0 = __double2loint(x[j].re), a1 = __double2hiint(x[j].re), a2 = __double2loint(x[j].im), a3 = __double2hiint(x[j].im);
        const int b0 = __builtin_amdgcn_ds_bpermute(srcl * 4, a0), b1 = __builtin_amdgcn_ds_bpermute(srcl * 4, a1);
        const int b2 = __builtin_amdgcn_ds_bpermute(srcl * 4, a2), b3 = __builtin_amdgcn_ds_bpermute(srcl * 4, a3);
        z[j] = cd{__hiloint2double(b1, b0), __hiloint2double(b3, b2)};
    }
    // new x[b] = z[(hi - b) & 7] = zr[(b - hi) & 7] with zr[j] = z[(8 - j) & 7]  -> rotate zr by (-hi) & 7
    cd zr[8];
#pragma unroll
    for (int j = 0; j < 8; j++) zr[j] = z[(8 - j) & 7];
    rot_by(zr, (8 - hi) & 7);
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = zr[j];
}

template <int MODE> __global__ __launch_bounds__(512) void k_x(double *out, int iters, int *bad)
{
    __shared__ cd sc[8][8 * 72];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    cd x[8]; double y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = cd{(double)(lane * 8 + i), (double)(1000 + lane * 8 + i)}; y[i] = lane + i; }
    if (MODE == 9) {
        cd a[8], b[8];
        for (int i = 0; i < 8; i++) a[i] = b[i] = x[i];
        xchg1_bperm(a, lane); xchg1_lds(b, sc[w], lane);
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
        if (MODE == 1 || MODE == 3) xchg1_bperm(x, lane);
        if (MODE == 4 || MODE == 2) xchg1_lds(x, sc[w], lane);
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) x[i].re += 1.0;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i].re + x[i].im + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> static void runx(const char *name, double *out, int *bad)
{
    const int iters = 4000;
    hipLaunchKernelGGL((k_x<MODE>), dim3(256), dim3(512), 0, 0, out, iters, bad); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL((k_x<MODE>), dim3(256), dim3(512), 0, 0, out, iters, bad); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-52s %.3f ms  (%.0f cycles per iteration per SIMD pair of waves @2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t *out; hipMalloc(&out, (size_t)cus * 4 * 256 * 8);
    printf("# part 1: DS instruction throughput per CU (cycles at a nominal 2.4 GHz)\n");
    rate<0>("ds_read_b32", 4, out, cus);
    rate<1>("ds_write_b32", 4, out, cus);
    rate<2>("ds_read_b128", 16, out, cus);
    rate<3>("ds_write_b128", 16, out, cus);
    rate<4>("ds_bpermute_b32 (lane*9 mod 64)", 4, out, cus);
    rate<7>("ds_bpermute_b32 (identity)", 4, out, cus);
    rate<5>("ds_permute_b32 (lane*9 mod 64)", 4, out, cus);
    rate<6>("ds_swizzle_b32 swizzle(SWAP,8)", 4, out, cus);
    printf("# part 2: exchange 1 of the radix-8 transform, 8 complex fp64 per lane, 8 waves per CU\n");
    int *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k_x<9>), dim3(4), dim3(512), 0, 0, (double *)out, 1, bad); hipDeviceSynchronize();
    int h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("bpermute exchange vs LDS exchange mismatches: %d\n", h);
    runx<0>("fp64 block only (96 fma)", (double *)out, bad);
    runx<1>("bpermute exchange only (32 bpermute + 2x96 selects)", (double *)out, bad);
    runx<4>("LDS exchange only (8 ds_write_b128 + 8 ds_read_b128)", (double *)out, bad);
    runx<3>("fp64 block + bpermute exchange", (double *)out, bad);
    runx<2>("fp64 block + LDS exchange", (double *)out, bad);
    return 0;
}
