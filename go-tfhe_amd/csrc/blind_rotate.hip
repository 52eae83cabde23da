// blind_rotate.hip -- instantiates and launches the blind-rotate / external-product kernels of the N = 1024 and N = 512 shapes
// (two-wave and four-wave forms) and dispatches the rest.  The fp64 kernels live in translation units of their own because the
// machine scheduler that suits them differs (build.py; measured per kernel family, profiles/r04_ab_scheduler_matrix.txt):
//   blind_rotate.hip        -mllvm -amdgpu-sched-strategy=max-ilp                        (-6 % at 1,024 gates against the default)
//   blind_rotate_oct.hip    the same + -mllvm -enable-post-misched=0                      (eight-wave kernel: -2.2 % at one bootstrap)
//   blind_rotate_n2048.hip  the default scheduler                                         (N = 2048, four-wave instances: -2.5 % at Uint5 x 512)
#include "launch_blind_rotate.hpp"


#include "kernels_n2048.hpp"      // for the extended-table instance only (launch_blind_rotate_ext2)
#include "kernels_n512.hpp"
#include "kernels_quad.hpp"

namespace tfhe {

void launch_blind_rotate(int shape, const BlindRotateArgs &a0, int B, int num_cus, int quad_limit, int oct_limit, hipStream_t st)
{
    // One launch covers at most the number of co-resident workgroups (28.8 KB LDS / 256 VGPRs -> 4 per CU
    // for N=1024; 55 KB LDS -> 2 per CU for N=2048).  All workgroups of such a launch walk the CMUX index in
    // near lock-step, so bsk[i] is shared through L2.  A single 16k-item launch de-synchronises (later
    // workgroups start as earlier ones finish) and becomes Infinity-Cache-bandwidth bound: 7.7 us per
    // bootstrap instead of 6.6; a persistent in-kernel item loop was tried and cost 9 % at B = 1024.
    // N=512: one wave and 16 KB LDS per bootstrap, 2 waves per SIMD -> 8 per CU.
    // N = 1024, launches of up to quad_max items: four waves per bootstrap (kernels_quad.hpp)
    const int quad_max = a0.bskq ? quad_limit : 0;
    const int cap = (shape_is_512(shape) ? 8 : shape_is_1024(shape) ? 4 : 2) * num_cus;
    for (int base = 0; base < B; base += cap) {
        const int cnt = B - base < cap ? B - base : cap;
        BlindRotateArgs a = a0;
        a.first = a0.first + base;
        a.batch = cnt;
        const dim3 g(cnt);
        if (shape_is_1024(shape) && cnt <= quad_max && cnt <= num_cus) {
            // at most one bootstrap per CU: eight waves per bootstrap where there are two or three gadget levels to split,
            // four otherwise (one wave per SIMD, all key levels prefetched).  The four-wave kernel at TWO workgroups per CU
            // (257...512 bootstraps) loses to the paired two-wave form below -- 4.31 vs 4.05 ms at 512, measured again in
            // round 3 with its twiddle loads scalar (profiles/r03_b_midsize.txt) -- and is no longer instantiated.
            if (cnt <= oct_limit && shape != kShapeN1024_L1_B23) {
                launch_blind_rotate_oct(shape, a, cnt, st);
            } else {
                switch (shape) {
                case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_blind_rotate_quad<3, 6, 1, 1>), g, dim3(256), 0, st, a); break;
                case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_blind_rotate_quad<2, 10, 1, 1>), g, dim3(256), 0, st, a); break;
                default: hipLaunchKernelGGL((k_blind_rotate_quad<1, 23, 1, 1>), g, dim3(256), 0, st, a); break;
                }
            }
            continue;
        }
        // Two items per four-wave workgroup (they share only the barriers), measured A/B on one box:
        //   769..1024 items: two workgroups on every CU, free-running, with the one-barrier step and the phase priorities
        //                    (FULL): 5.45 ms at 1,024 against 5.93 for FOUR items in one eight-wave workgroup in step on
        //                    the key stream (rounds 1-3), 6.27 for the same two workgroups without priorities;
        //   257..512  items: one such workgroup per CU: the hardware leaves a SIMD idle with two 2-wave
        //                    workgroups per CU (see k_blind_rotate), 5.15 -> 4.42 ms at 512;
        //   513..768  items: one item per workgroup (pairing measured 6.40 vs 5.65 ms at 768), with the phase priorities:
        //                    5.26 -> 4.57 ms at 768.
        if (shape_is_1024(shape) && cnt > 3 * num_cus) {
            const dim3 g2((cnt + 1) / 2);
            switch (shape) {
            case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_blind_rotate<3, 6, 2, true>), g2, dim3(256), 0, st, a); break;
            case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_blind_rotate<2, 10, 2, true>), g2, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((k_blind_rotate<1, 23, 2, true>), g2, dim3(256), 0, st, a); break;
            }
            continue;
        }
        if (shape_is_1024(shape) && cnt > num_cus && cnt <= 2 * num_cus) {
            const dim3 g2((cnt + 1) / 2);
            switch (shape) {
            case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_blind_rotate<3, 6, 2>), g2, dim3(256), 0, st, a); break;
            case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_blind_rotate<2, 10, 2>), g2, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((k_blind_rotate<1, 23, 2>), g2, dim3(256), 0, st, a); break;
            }
            continue;
        }
        if (shape_is_1024(shape) && cnt > 2 * num_cus) {        // 513..768: three two-wave workgroups per CU, phase priorities
            switch (shape) {
            case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_blind_rotate<3, 6, 1, true>), g, dim3(128), 0, st, a); break;
            case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_blind_rotate<2, 10, 1, true>), g, dim3(128), 0, st, a); break;
            default: hipLaunchKernelGGL((k_blind_rotate<1, 23, 1, true>), g, dim3(128), 0, st, a); break;
            }
            continue;
        }
        switch (shape) {
        case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_blind_rotate<3, 6>), g, dim3(128), 0, st, a); break;
        case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_blind_rotate<2, 10>), g, dim3(128), 0, st, a); break;
        case kShapeN1024_L1_B23: hipLaunchKernelGGL((k_blind_rotate<1, 23>), g, dim3(128), 0, st, a); break;
        case kShapeN512_L1_B18: hipLaunchKernelGGL((k_blind_rotate_512<18>), g, dim3(64), 0, st, a); break;
        default: launch_blind_rotate_2048(a, cnt, num_cus, st); break;
        }
    }
}

// The extended-table form of the N = 2048 kernel (EXT = 2: eight waves, one workgroup per CU) is scheduled like the N = 1024 kernels, not like
// its four-wave siblings: Uint6 x 64 5.52 ms here against 5.87 ms in the default-scheduled unit (tools/ext_bench.py, three rounds on one box).
void launch_blind_rotate_ext2(const BlindRotateArgs &a0, int B, hipStream_t st)
{
    BlindRotateArgs a = a0;
    a.batch = B;
    hipLaunchKernelGGL((k_blind_rotate_2048<22, false, 2>), dim3(B), dim3(512), 0, st, a);
}

void launch_external_product(int shape, const cd *bsk, const cd *tw, int key_index, const uint32_t *in, uint32_t *out,
                             uint32_t offset, int B, hipStream_t st)
{
    switch (shape) {
    case kShapeN1024_L3_B6: hipLaunchKernelGGL((k_external_product<3, 6>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset); break;
    case kShapeN1024_L2_B10: hipLaunchKernelGGL((k_external_product<2, 10>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset); break;
    case kShapeN1024_L1_B23: hipLaunchKernelGGL((k_external_product<1, 23>), dim3(B), dim3(128), 0, st, bsk, tw, key_index, in, out, offset); break;
    case kShapeN512_L1_B18: hipLaunchKernelGGL((k_external_product_512<18>), dim3(B), dim3(64), 0, st, bsk, tw, key_index, in, out, offset); break;
    default: launch_external_product_2048(bsk, tw, key_index, in, out, offset, B, st); break;
    }
}

} // namespace tfhe
