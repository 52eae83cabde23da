// blind_rotate.hip -- instantiates and launches the blind-rotate / external-product kernels.
// Separate translation unit: built with -mllvm -amdgpu-sched-strategy=max-ilp (see build.py).
#include "launch_blind_rotate.hpp"

#include "kernels_n2048.hpp"

namespace tfhe {

void launch_blind_rotate(int shape, const BlindRotateArgs &a, int B, hipStream_t st)
{
    switch (shape) {
    case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_blind_rotate<3, 6>), dim3(B), dim3(128), 0, st, a); break;
    case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_blind_rotate<2, 10>), dim3(B), dim3(128), 0, st, a); break;
    case kShapeN1024_L1_B23: hipLaunchKernelGGL((k_blind_rotate<1, 23>), dim3(B), dim3(128), 0, st, a); break;
    default: hipLaunchKernelGGL((k_blind_rotate_2048<22>), dim3(B), dim3(256), 0, st, a); break;
    }
}

void launch_external_product(int shape, const cd *bsk, const cd *tw, int key_index, const uint32_t *in, uint32_t *out,
                             uint32_t offset, int B, hipStream_t st)
{
    switch (shape) {
    case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_external_product<3, 6>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset); break;
    case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_external_product<2, 10>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset); break;
    case kShapeN1024_L1_B23: hipLaunchKernelGGL((k_external_product<1, 23>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset); break;
    default: hipLaunchKernelGGL((k_external_product_2048<22>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset); break;
    }
}

} // namespace tfhe
