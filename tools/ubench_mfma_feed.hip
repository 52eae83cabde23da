// What a 16-MFMA chunk of k_keyswitch_mfma costs as its feeding is added piece by piece (one workgroup of 4 waves per CU):
//   A: 16 independent v_mfma_i32_32x32x32_i8          B: + its 8 operand ds_read_b128
//   C: + per two chunks 8 ds_write_b128 and a barrier  D: + per two chunks 8 global_load_dwordx4 (a 142 MB working set)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_feed.hip -o /tmp/ubench_mfma_feed && /tmp/ubench_mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void loop(int iters, const uint4 *__restrict__ g, size_t gmask, int *sink)
{
    __shared__ uint4 lds[2][8][256];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    v16i acc[4][4];
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) for (int r = 0; r < 16; r++) acc[a][b][r] = 0;
    for (int q = 0; q < 8; q++) lds[0][q][tid] = lds[1][q][tid] = make_uint4(tid, q, 1, 2);
    __syncthreads();
    v4i a[4], b[4];
    for (int q = 0; q < 4; q++) { a[q] = (v4i){tid, q, 1, 2}; b[q] = (v4i){3, tid, q, 4}; }
    size_t off = (size_t)blockIdx.x * 65536 + tid;
    uint4 r0 = g[off & gmask], r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = r0, r7 = r0;
    for (int i = 0; i < iters; i++) {
        const int buf = i & 1;
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            if (MODE >= 1) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    __builtin_memcpy(&a[q], &lds[buf][2 * ch + (l >> 5)][wm * 128 + 32 * q + (l & 31)], 16);
                    __builtin_memcpy(&b[q], &lds[buf][4 + 2 * ch + (l >> 5)][wn * 128 + 32 * q + (l & 31)], 16);
                }
            }
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (MODE >= 2) {
            lds[buf ^ 1][0][tid] = r0; lds[buf ^ 1][1][tid] = r1; lds[buf ^ 1][2][tid] = r2; lds[buf ^ 1][3][tid] = r3;
            lds[buf ^ 1][4][tid] = r4; lds[buf ^ 1][5][tid] = r5; lds[buf ^ 1][6][tid] = r6; lds[buf ^ 1][7][tid] = r7;
        }
        if (MODE >= 3) {
            off += 4096 * 37;
            r0 = g[(off) & gmask]; r1 = g[(off + 16384) & gmask]; r2 = g[(off + 2 * 16384) & gmask]; r3 = g[(off + 3 * 16384) & gmask];
            r4 = g[(off + 4 * 16384) & gmask]; r5 = g[(off + 5 * 16384) & gmask]; r6 = g[(off + 6 * 16384) & gmask]; r7 = g[(off + 7 * 16384) & gmask];
        }
        if (MODE >= 2) __syncthreads();
    }
    int s = r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
    for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) for (int r = 0; r < 16; r++) s += acc[x][y][r];
    if (s == 0x12345678) sink[0] = s;
}
template <int MODE> void run(const char *what, const uint4 *g, size_t gmask, int *sink)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, wgs = 256;
    hipLaunchKernelGGL(loop<MODE>, dim3(wgs), dim3(256), 0, 0, 8, g, gmask, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(loop<MODE>, dim3(wgs), dim3(256), 0, 0, iters, g, gmask, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %7.3f ms  %6.0f cycles per 32-MFMA stage @2.4GHz\n", what, ms, ms * 1e-3 * 2.4e9 / iters);
}
int main()
{
    const size_t elems = (size_t)1 << 23;       // 8 M x 16 B = 128 MB
    uint4 *g; int *sink; (void)hipMalloc(&g, elems * 16); (void)hipMalloc(&sink, 4); (void)hipMemset(g, 1, elems * 16);
    run<0>("A: 32 MFMA", g, elems - 1, sink);
    run<1>("B: + 16 operand ds_read_b128", g, elems - 1, sink);
    run<2>("C: + 8 ds_write_b128 + barrier", g, elems - 1, sink);
    run<3>("D: + 8 global_load_dwordx4 (128 MB set)", g, elems - 1, sink);
    return 0;
}
