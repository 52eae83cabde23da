// keyswitch_mfma.hpp -- the base-4 identity key switch (keyswitch.go:10-37) as an exact int8 matrix product.
//
// out[m] = (0, ..., 0, b_m) - sum_{i < N, j < t} KSK[(i t + j) 4 + d(m, i, j)], d = the j-th base-4 digit of the
// i-th extracted coefficient.  Written over all four candidate rows of a digit (row k = 0 is all-zero,
// keyswitch.go:30) that is  out = init - H x KSK  with H[m][K] in {0, 1} one-hot per digit, K = N t 4 = 36,864 at
// the 128-bit set: 1,024 x 36,864 x 2,804 byte-columns per batch.  The 32-bit words of the key are split into their four
// bytes, each byte column summed exactly by v_mfma_i32_32x32x32_i8 (at most N t = 9,216 terms of magnitude <= 128:
// 21 bits), and the four column sums of a word recombined mod 2^32 -- bit-identical to the row-by-row subtraction.
// The instruction multiplies SIGNED bytes, so the key copy holds u ^ 0x80 (= u - 128); every digit selects exactly
// one row, so the correction is the constant 128 N t per byte column.
//
// Layouts (both operands are stored as the instruction's lanes read them: one 16-byte piece = 16 consecutive K of
// one row / column):
//   K order   K = (j N + i) base + k: piece kb = j N/4 + i/4 holds coefficients 4 (i/4) .. +3 of digit level j (base 4);
//             with base 16 (Uint2) a piece is one coefficient's 16 candidates, kb = j N + i
//   kskB      int8 [t N / 4][colsP][16]   colsP = 4 (n + 1) rounded up to 256; built once per key (k_ksk_mfma_pack)
//   H         int8 [t N / 4][Mpad][16]    Mpad = ciphertexts rounded up to 256; built per launch (k_ks_onehot):
//                                         a digit is the 32-bit word 1 << 8 d
// One wave = one 64 x 128 tile of (ciphertexts x byte columns) over a range of K: 2 + 4 operand pieces of 16 bytes per
// lane feed 8 MFMAs; partial sums over the K ranges are combined with 32-bit atomics into `out`, which k_ks_init has
// set to (0, ..., 0, b).  Lane layout of the instruction: tools/probe_mfma_i8.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe {

typedef int ks_v4i __attribute__((ext_vector_type(4)));
typedef int ks_v16i __attribute__((ext_vector_type(16)));


// packed key [N][t][base - 1][n1p] (k = 1 .. base - 1; kernels.hpp) -> kskB.  BB = basebit: 2 (four coefficients per
// 16-K piece) or 4 (one coefficient per piece).
template <int BB>
static __global__ void k_ksk_mfma_pack(const uint32_t *__restrict__ packed, uint4 *__restrict__ dst, int N, int t, int n1,
                                       int n1p, int colsP)
{
    constexpr int base = 1 << BB, DPP = 16 / base;          // digits per piece
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)t * (N / DPP) * colsP;
    if (idx >= total) return;
    const int col = (int)(idx % colsP), kb = (int)(idx / colsP);
    const int j = kb / (N / DPP), i0 = DPP * (kb % (N / DPP));
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 16; b++) {                          // byte b of the piece: coefficient i0 + b / base, candidate k = b % base
        const int ii = b / base, k = b % base;
        uint32_t u = 0;
        if (k > 0 && col < 4 * n1) {
            const uint32_t word = packed[((size_t)((i0 + ii) * t + j) * (base - 1) + (k - 1)) * n1p + (col >> 2)];
            u = (word >> (8 * (col & 3))) & 0xFFu;
        }
        w[b >> 2] |= (u ^ 0x80u) << (8 * (b & 3));
    }
    dst[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Sample extract at index 0 (trlwe_ops.go:13-19) + digit decomposition (keyswitch.go:14-16,25-29) -> H.
// Workgroup = 16 ciphertexts x 16 coefficient quads; stores of one digit level are 256 contiguous bytes per piece.
template <int BB>
static __global__ __launch_bounds__(256) void k_ks_onehot(const uint32_t *__restrict__ trlwe, uint4 *__restrict__ H, int N, int t,
                                                          int M, int Mpad, const int *__restrict__ count, int m_base)
{
    constexpr int base = 1 << BB, DPP = 16 / base;
    const int tid = threadIdx.x, m = 16 * blockIdx.x + (tid & 15), iq = 16 * blockIdx.y + (tid >> 4);
    int live_items = M;
    if (count) { const int c = *count - m_base; live_items = c < M ? (c < 0 ? 0 : c) : M; }
    // rows past the live items stay as they are: the rows of a matrix product are independent and k_keyswitch_mfma
    // never stores theirs
    if (m >= live_items) return;
    const uint32_t prec = 1u << (32 - (1 + BB * t));
    uint32_t w[4];
    const uint32_t *ta = trlwe + (size_t)m * 2 * N;
#pragma unroll
    for (int ii = 0; ii < 4; ii++) {
        const int i = 4 * iq + ii;
        const uint32_t ai = i == 0 ? ta[0] : ~ta[N - i];
        w[ii] = (ai + prec) >> (32 - BB * t);                 // t digits, most significant first
    }
    for (int j = 0; j < t; j++) {
        const int sh = BB * (t - 1 - j);
        if (DPP == 4) {                                       // a digit is the word 1 << 8 d
            H[((size_t)j * (N / 4) + iq) * Mpad + m] = make_uint4(1u << (8 * ((w[0] >> sh) & 3)), 1u << (8 * ((w[1] >> sh) & 3)),
                                                                  1u << (8 * ((w[2] >> sh) & 3)), 1u << (8 * ((w[3] >> sh) & 3)));
        } else {                                              // a digit is a whole piece: byte d of 16 set
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                const uint32_t d = (w[ii] >> sh) & (base - 1), bit = 1u << (8 * (d & 3));
                H[((size_t)j * N + 4 * iq + ii) * Mpad + m] = make_uint4((d >> 2) == 0 ? bit : 0u, (d >> 2) == 1 ? bit : 0u,
                                                                         (d >> 2) == 2 ? bit : 0u, (d >> 2) == 3 ? bit : 0u);
            }
        }
    }
}

// One workgroup = eight waves (two per SIMD) = a 256 x 256 tile of (ciphertexts x byte columns) over a range of K, wave
// (wm, wn) owning 64 rows x 128 columns = 8 accumulator tiles (128 registers).  K advances in stages of kKsStage chunks
// of 32.  Every thread moves two 16-byte pieces of H and two of kskB per stage: global memory -> registers three stages
// ahead (two alternating register sets), registers -> the other LDS buffer one stage ahead -- those stores are issued
// BEFORE the stage's MFMAs so that they complete under them -- one barrier per stage; operands reach the instruction
// through ds_read_b128.  Each operand piece crosses L2 -> CU once per workgroup and is read from LDS by the waves that
// need it.  The launch is sized to at most one workgroup per CU, with as many K ranges as that allows (uneven, never a
// second partial round).
// Measured alternatives (profiles/r02_h_ks_pmc.txt): four waves of 128 x 128 (one per SIMD) 0.134 ms at 1,024 ciphertexts
// against 0.120; operands loaded straight into LDS (global_load_lds_dwordx4, three-slot ring, bare s_barrier) 0.126;
// four chunks per stage 0.144; one-stage look-ahead 0.27.
constexpr int kKsStage = 2;                  // chunks of 32 K per stage: 2 x 32 KB of LDS
constexpr int kKsGroup = 256;                // workgroup tile edge

__global__ __launch_bounds__(512) void k_keyswitch_mfma(const uint4 *__restrict__ H, const uint4 *__restrict__ kskB,
                                                        uint32_t *__restrict__ out, int Mpad, int colsP, int n1, int M,
                                                        const int *__restrict__ count, int m_base, int m_groups, int n_groups,
                                                        int pairs_total, int pairs_per_part, int parts, uint32_t bias_word)
{
    constexpr int P = 2 * kKsStage;          // 16-K pieces per stage
    constexpr int MI = 2;                    // 32-row accumulator tiles per wave (x 4 column tiles)
    static_assert(P == 4, "the staging registers below are written out for two chunks per stage");
    __shared__ uint4 ldsA[2][P][kKsGroup], ldsB[2][P][kKsGroup];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int t8 = tid & 255, half = tid >> 8;                   // staging: thread t8 of half `half` moves pieces 2 half, 2 half + 1
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): XCD x takes the work items
    // [x S, x S + S) in the order (K range, column group, ciphertext group), so the workgroups that run side by side on an
    // XCD share their kskB pieces (same K range and column group) and mostly their H pieces (same K range).
    const int per_xcd = (int)(gridDim.x >> 3), wi = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (wi >= m_groups * n_groups * parts) return;
    const int mg = wi % m_groups, ng = (wi / m_groups) % n_groups, kp = wi / (m_groups * n_groups);
    const int m0 = mg * kKsGroup, n0 = ng * kKsGroup;
    int live_items = M;
    if (count) { const int c = *count - m_base; live_items = c < M ? (c < 0 ? 0 : c) : M; }
    if (m0 >= live_items) return;
    const bool rows_live = m0 + wm * 32 * MI < live_items;      // (a wave whose rows are all padding multiplies garbage, stores nothing)
    const int pair0 = kp * pairs_per_part;
    const int pairs = pairs_total - pair0 < pairs_per_part ? pairs_total - pair0 : pairs_per_part;     // stage pairs of this range
    if (pairs <= 0) return;                                      // (uneven split: a trailing range may be empty)
    const int stages = 2 * pairs;
    const uint4 *gA = H + ((size_t)pair0 * 2 * P + 2 * half) * Mpad + m0 + t8;
    const uint4 *gB = kskB + ((size_t)pair0 * 2 * P + 2 * half) * colsP + n0 + t8;
    const size_t stepA = (size_t)P * Mpad, stepB = (size_t)P * colsP;

    ks_v16i acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0;

    // stage index -> its pieces (clamped to the range: the tail re-reads the last stage, never used).  Macros over named
    // registers, not lambdas over arrays: register arrays passed by reference end up in scratch memory.
#define KS_FETCH(st, R)                                                                                   \
    {                                                                                                     \
        const int sc_ = (st) < stages ? (st) : stages - 1;                                                \
        const uint4 *pa_ = gA + (size_t)sc_ * stepA, *pb_ = gB + (size_t)sc_ * stepB;                     \
        R##a0 = pa_[0]; R##a1 = pa_[(size_t)Mpad]; R##b0 = pb_[0]; R##b1 = pb_[(size_t)colsP];            \
    }
#define KS_STASH(buf, R)                                                                                  \
    {                                                                                                     \
        ldsA[buf][2 * half][t8] = R##a0; ldsA[buf][2 * half + 1][t8] = R##a1;                             \
        ldsB[buf][2 * half][t8] = R##b0; ldsB[buf][2 * half + 1][t8] = R##b1;                             \
    }
#define KS_MULTIPLY(buf)                                                                                  \
    _Pragma("unroll") for (int ch = 0; ch < kKsStage; ch++) {                                             \
        ks_v4i a[MI], b[4];                                                                               \
        _Pragma("unroll") for (int q = 0; q < MI; q++)                                                    \
            __builtin_memcpy(&a[q], &ldsA[buf][2 * ch + (l >> 5)][wm * 32 * MI + 32 * q + (l & 31)], 16); \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                                     \
            __builtin_memcpy(&b[q], &ldsB[buf][2 * ch + (l >> 5)][wn * 128 + 32 * q + (l & 31)], 16);    \
        _Pragma("unroll") for (int mi = 0; mi < MI; mi++)                                                 \
            _Pragma("unroll") for (int ni = 0; ni < 4; ni++)                                              \
                acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], acc[mi][ni], 0, 0, 0);  \
    }

    uint4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
    KS_FETCH(0, x);
    KS_STASH(0, x);
    KS_FETCH(1, x);             // set x: stage s + 1 at the top of an even stage s
    KS_FETCH(2, y);             // set y: stage s + 2
    __syncthreads();
    for (int s = 0; s < stages; s += 2) {
        // the stores to the other buffer go out FIRST and complete under the MFMAs; after them they cost ~1,000 cycles
        // per stage (tools/ubench_mfma_feed.hip: 32 MFMA 1,240 cycles, + operand reads 1,460, + stores and barrier 2,510)
        KS_STASH(1, x);
        __builtin_amdgcn_sched_barrier(0);
        KS_MULTIPLY(0);
        KS_FETCH(s + 3, x);
        __syncthreads();
        KS_STASH(0, y);
        __builtin_amdgcn_sched_barrier(0);
        KS_MULTIPLY(1);
        KS_FETCH(s + 4, y);
        __syncthreads();
    }
#undef KS_FETCH
#undef KS_STASH
#undef KS_MULTIPLY
    if (!rows_live) return;

    // D[row (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col l & 31]: the four byte columns of an output word sit in four
    // adjacent lanes; shift each to its place, add across the quad, lane 0 of the quad subtracts from `out`
    const uint32_t fix = kp == 0 ? bias_word : 0u;
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++) {
            const int word = ((n0 + wn * 128 + 32 * ni) >> 2) + ((l & 31) >> 2);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                uint32_t v = (uint32_t)acc[mi][ni][r] << (8 * (l & 3));
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
                v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
                const int row = m0 + wm * 32 * MI + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if ((l & 3) == 0 && word < n1 && row < live_items) atomicAdd(&out[(size_t)row * n1 + word], 0u - (v + fix));
            }
        }
}

} // namespace tfhe
