// blind_rotate_oct.hip -- the eight-wave blind rotate of the N = 1024 shapes (kernels_quad.hpp: launches of at most one bootstrap per
// CU).  A translation unit of its own: max-ILP machine scheduler WITHOUT the post-register-allocation scheduling pass
// (-mllvm -enable-post-misched=0), which reorders the kernel's serial tail for the worse: 2.195 -> 2.147 ms at one bootstrap on two boxes,
// +-0 at 256; the other kernels are indifferent to it or a little slower (profiles/r04_ab_scheduler_matrix.txt; build.py).
#include "launch_blind_rotate.hpp"

#include "kernels_quad.hpp"

namespace tfhe {

void launch_blind_rotate_oct(int shape, const BlindRotateArgs &a, int cnt, hipStream_t st)
{
    const dim3 g(cnt);
    if (shape == kShapeN1024_L3_B6) hipLaunchKernelGGL((k_blind_rotate_oct<3, 6>), g, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_blind_rotate_oct<2, 10>), g, dim3(512), 0, st, a);
}

} // namespace tfhe
