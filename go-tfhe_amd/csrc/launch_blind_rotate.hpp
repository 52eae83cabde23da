// launch_blind_rotate.hpp -- the fp64-heavy kernels live in their own translation unit
// (blind_rotate.hip) so they can be compiled with the max-ILP machine scheduler
// (-mllvm -amdgpu-sched-strategy=max-ilp: -5 % blind-rotate time, but +35 % on the memory-bound
// key-switch kernels, which therefore stay in the default-scheduled unit).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace tfhe {
// shape 1: N=1024, L=3, Bgbit=6   shape 2: N=2048, L=1, Bgbit=22
void launch_blind_rotate(int shape, const BlindRotateArgs &args, int B, hipStream_t st);
void launch_external_product(int shape, const cd *bsk, const cd *tw, int key_index, const uint32_t *in, uint32_t *out,
                             uint32_t offset, int B, hipStream_t st);
} // namespace tfhe
