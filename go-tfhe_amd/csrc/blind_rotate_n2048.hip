// blind_rotate_n2048.hip -- the blind-rotate / external-product kernels of the N = 2048 shape (kernels_n2048.hpp).  A translation
// unit of its own, built with the DEFAULT machine scheduler: the max-ILP strategy that buys the N = 1024 kernels 6 % costs these 2.5 %
// (Uint5 x 512: 5.40 / 5.33 ms against 5.23 / 5.17 on two boxes, interleaved; profiles/r04_ab_scheduler_matrix.txt; build.py).
// The extended-table instance (EXT = 2, eight waves) is the exception -- 5.52 ms with max-ILP against 5.87 here -- and lives in blind_rotate.hip.
#include "launch_blind_rotate.hpp"

#include "kernels_n2048.hpp"

namespace tfhe {

void launch_blind_rotate_2048(const BlindRotateArgs &a, int cnt, int num_cus, hipStream_t st)
{
    // one bootstrap per four-wave workgroup whatever the launch size (two bootstraps per eight-wave workgroup
    // were 2 % faster while a step had five barriers; with four, free-running workgroups win by 7 %)
    const dim3 g(cnt);
    if (cnt <= num_cus) hipLaunchKernelGGL((k_blind_rotate_2048<22, true>), g, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_blind_rotate_2048<22, false>), g, dim3(256), 0, st, a);
}

void launch_external_product_2048(const cd *bsk, const cd *tw, int key_index, const uint32_t *in, uint32_t *out, uint32_t offset, int B,
                                  hipStream_t st)
{
    hipLaunchKernelGGL((k_external_product_2048<22>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset);
}

} // namespace tfhe
