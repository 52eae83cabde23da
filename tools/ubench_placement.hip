// Where does the hardware place the waves of small grids?  Each wave records HW_ID (simd, cu, se) and XCC_ID
// while holding the resource footprint of k_blind_rotate (256 VGPRs, 28.8 KB LDS, 128 threads), spinning
// long enough for the whole grid to be co-resident.   hipcc --offload-arch=gfx950 -O2 tools/ubench_placement.hip -o /tmp/placement && /tmp/placement 512
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(128, 2) void k_probe(unsigned *out, long spin)
{
    __shared__ char lds[28800];
    lds[threadIdx.x] = 0;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");                         // force a 256-VGPR allocation
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));      // HW_REG_XCC_ID
    const long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 2 + (threadIdx.x >> 6);
        out[2 * w] = hw; out[2 * w + 1] = xcc + (unsigned)lds[0];
    }
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 512;
    unsigned *d; hipMalloc(&d, (size_t)B * 2 * 2 * 4);
    hipLaunchKernelGGL(k_probe, dim3(B), dim3(128), 0, 0, d, 20000000L);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)B * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, cu): waves on each SIMD
    std::map<unsigned, std::vector<int>> cu;
    for (int w = 0; w < 2 * B; w++) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xF;
        const unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        auto &v = cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid];
        if (v.empty()) v.assign(4, 0);
        v[simd]++;
    }
    std::map<std::vector<int>, int> hist;
    for (auto &kv : cu) { auto v = kv.second; hist[v]++; }
    printf("B=%d workgroups x 2 waves: %zu CUs used; per-CU waves on SIMD0..3 -> number of CUs\n", B, cu.size());
    for (auto &kv : hist) printf("  [%d %d %d %d] : %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    // do the two waves of a workgroup sit on different SIMDs?
    int same = 0;
    for (int b = 0; b < B; b++) same += ((h[4 * b] >> 4) & 3) == ((h[4 * b + 2] >> 4) & 3);
    printf("workgroups with both waves on one SIMD: %d of %d\n", same, B);
    return 0;
}
