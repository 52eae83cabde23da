// blind_rotate.hip -- instantiates and launches the blind-rotate / external-product kernels.
// Separate translation unit: built with -mllvm -amdgpu-sched-strategy=max-ilp (see build.py).
#include "launch_blind_rotate.hpp"

#include "kernels_n2048.hpp"

namespace tfhe {

void launch_blind_rotate(int shape, const BlindRotateArgs &a, int B, hipStream_t st)
{
    if (shape == 1)
        hipLaunchKernelGGL((k_blind_rotate<3, 6>), dim3(B), dim3(128), 0, st, a);
    else
        hipLaunchKernelGGL((k_blind_rotate_2048<22>), dim3(B), dim3(256), 0, st, a);
}

void launch_external_product(int shape, const cd *bsk, const cd *tw, int key_index, const uint32_t *in, uint32_t *out,
                             uint32_t offset, int B, hipStream_t st)
{
    if (shape == 1)
        hipLaunchKernelGGL((k_external_product<3, 6>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset);
    else
        hipLaunchKernelGGL((k_external_product_2048<22>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset);
}

} // namespace tfhe
