// launch_blind_rotate.hpp -- the fp64-heavy kernels live in their own translation units
// (blind_rotate*.hip) so they can be compiled with the machine-scheduler options that suit them
// (max-ILP: -6 % blind-rotate time at N = 1024, but +35 % on the memory-bound key-switch kernels,
// which therefore stay in the default-scheduled unit, and +2.5 % on the N = 2048 kernels).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace tfhe {
// Parameter shapes with kernels (tfhe_ctx_create picks one):
enum Shape { kShapeN1024_L3_B6 = 1,   // 80/110/128-bit sets
             kShapeN2048_L1_B22 = 2,  // Uint4, Uint5, Uint6 (and the shapes of Uint7/8)
             kShapeN1024_L2_B10 = 3,  // Uint1
             kShapeN1024_L1_B23 = 4,  // Uint3
             kShapeN512_L1_B18 = 5 }; // Uint2
inline bool shape_is_1024(int shape) { return shape != kShapeN2048_L1_B22 && shape != kShapeN512_L1_B18; }
inline bool shape_is_512(int shape) { return shape == kShapeN512_L1_B18; }
// B items in launches of at most the co-resident workgroup count (4 per CU for N=1024, 2 per CU for N=2048).
// quad_limit: N = 1024 launches of up to this many items use the four-wave kernel (kernels_quad.hpp);
// oct_limit: of those, launches of up to this many (and at most one per CU) use the eight-wave kernel.
void launch_blind_rotate(int shape, const BlindRotateArgs &args, int B, int num_cus, int quad_limit, int oct_limit, hipStream_t st);
// The pieces of launch_blind_rotate / launch_external_product that live in translation units of their own (blind_rotate_oct.hip,
// blind_rotate_n2048.hip: other machine-scheduler options, see blind_rotate.hip): one launch of `cnt` items, args already offset.
void launch_blind_rotate_oct(int shape, const BlindRotateArgs &args, int cnt, hipStream_t st);
void launch_blind_rotate_2048(const BlindRotateArgs &args, int cnt, int num_cus, hipStream_t st);
void launch_external_product_2048(const cd *bsk, const cd *tw, int key_index, const uint32_t *in, uint32_t *out, uint32_t offset, int B,
                                  hipStream_t st);
// Persistent blind rotate through an extended lookup table with polyExtendFactor 2 (kernels_n2048.hpp, EXT = 2): one
// eight-wave workgroup per item; args.tv = lut [2][2][N] (tv_stride 0 or 4N), args.in1 / ops / idx unused.
void launch_blind_rotate_ext2(const BlindRotateArgs &args, int B, hipStream_t st);
void launch_external_product(int shape, const cd *bsk, const cd *tw, int key_index, const uint32_t *in, uint32_t *out,
                             uint32_t offset, int B, hipStream_t st);
} // namespace tfhe
