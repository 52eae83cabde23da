// kernels_n512.hpp -- the N = 512 shape (Uint2: L = 1, Bgbit = 18; params.go:236-265).
//
// M = N/2 = 256 complex points are half of what one wavefront holds at 8 points per lane, so a
// wavefront carries TWO independent 256-point transforms, one per 32-lane half.  With L = 1 that is
// exactly one CMUX step's worth: half 0 transforms the digit polynomial of accumulator A, half 1
// that of B; after the pointwise products the cross-half sum of the external product
// (evaluator.go:59-61: out = dA*row0 + dB*row1) is one v_permlane32_swap per dword and an add, and
// the same wave inverse-transforms A in half 0 and B in half 1.  One bootstrap = one wavefront = one
// workgroup: no partner wave, no s_barrier anywhere in the CMUX loop.
//
// The 256-point tree has the structure of fft512_forward with a radix-4 last level:
//
//   level 1 (a -> m)   exchange   level 2 (b -> m')   exchange   level 3 (c -> m''), two per lane
//   j = 32a + 4b + c   reg a, hl 4b+c  -->  reg b, hl 4m+c  -->  reg 4q+c, hl 4m+i   (m' = 4q+i)
//
// hl = lane & 31; the value left in (reg 4q+m'', hl 4m+i) is Z(zeta^(1+4u)), zeta = exp(i pi/512),
// u = m + 8m' + 64m''.
#pragma once

#include "kernels.hpp"

namespace tfhe {

// Twiddle table for N = 512 (built on the host in long double):
//   [0..7]                    c1[a] = zeta^(32 a)                     (uniform)
//   [8..15]                   conj(c1[a]) / 256
//   [16 + b*32 + hl]          c2 = zeta^(4 b (1+4m)),                 m = hl>>2
//   [272 + (4q+c)*32 + hl]    c3 = zeta^(c (1+4(m+8m'))),             m' = 4q + (hl&3)
constexpr int kTw512Level2 = 16;
constexpr int kTw512Level3 = 16 + 256;
constexpr int kTwCount512 = 16 + 256 + 256;
// per-half exchange scratch: 8 rows of 36 slots (32 used + 4 pad: row stride = 4 mod 16 keeps the
// 16-lane ds_read_b128 service groups on distinct banks); two halves = kScratchSlots
constexpr int kHalfScratch = 8 * 36;
constexpr int kScratchSlots512 = 2 * kHalfScratch;      // (= the padded kScratchSlots of negacyclic_fft.hpp)

struct LaneTwiddles512 {
    TwPow l2;        // level 2: w, w^2, w^4
    cd v[2][3];      // level 3: w_q, w_q^2, w_q^3 for the lane's two radix-4 butterflies
};

__device__ __forceinline__ void load_lane_twiddles_512(LaneTwiddles512 &tw, const cd *__restrict__ table, int hl)
{
    tw.l2.w1 = table[kTw512Level2 + 1 * 32 + hl];
    tw.l2.w2 = table[kTw512Level2 + 2 * 32 + hl];
    tw.l2.w4 = table[kTw512Level2 + 4 * 32 + hl];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 1; c < 4; c++) tw.v[q][c - 1] = table[kTw512Level3 + (4 * q + c) * 32 + hl];
}

__host__ __device__ __forceinline__ int spectrum_u_512(int reg, int hl)
{
    return (hl >> 2) + 8 * (4 * (reg >> 2) + (hl & 3)) + 64 * (reg & 3);
}

// Slot in the reference FourierPoly order: s = bitrev8(-u mod 256) (same derivation as
// reference_slot_1024, one bit shorter).
__host__ __device__ __forceinline__ int reference_slot_512(int reg, int hl)
{
    int v = (256 - spectrum_u_512(reg, hl)) & 255, s = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) s |= ((v >> b) & 1) << (7 - b);
    return s;
}

// y_k = sum_n x_n exp(S * 2 pi i n k / 4)
template <int S> __device__ __forceinline__ void dft4(cd &x0, cd &x1, cd &x2, cd &x3)
{
    const cd b0 = x0 + x2, b1 = x1 + x3, d0 = x0 - x2, d1 = mul_i<S>(x1 - x3);
    x0 = b0 + b1; x2 = b0 - b1; x1 = d0 + d1; x3 = d0 - d1;
}

// Forward transform of 256 complex points held by one half-wave; x[a] = z_{32a+hl} in, spectrum
// order out.  sch = this half's kHalfScratch slots.
__device__ __forceinline__ void fft256_forward(cd (&x)[8], cd *sch, const cd *__restrict__ table,
                                               const LaneTwiddles512 &tw, int hl)
{
    const int hi = hl >> 2, lo = hl & 3;
    dft8_pretwist<1>(x, table[1], table[2], table[3], table[4], table[5], table[6], table[7]);
    // exchange 1: (reg m, hl 4b+c) -> (reg b, hl 4m+c)
#pragma unroll
    for (int m = 0; m < 8; m++) sch[36 * m + hl] = x[m];
    wave_lds_order();
#pragma unroll
    for (int b = 0; b < 8; b++) x[b] = sch[36 * hi + 4 * b + lo];
    wave_lds_order();
    twist_pow_dft8(x, tw.l2);
    // exchange 2: (reg m', hl 4m+c) -> (reg 4q+c, hl 4m+i), m' = 4q+i; slot 36m + 4m' + ((c+m')&3)
#pragma unroll
    for (int mp = 0; mp < 8; mp++) sch[36 * hi + 4 * mp + ((lo + mp) & 3)] = x[mp];
    wave_lds_order();
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 4; c++) x[4 * q + c] = sch[36 * hi + 4 * (4 * q + lo) + ((c + lo) & 3)];
    wave_lds_order();
#pragma unroll
    for (int q = 0; q < 2; q++) {
#pragma unroll
        for (int c = 1; c < 4; c++) x[4 * q + c] = cmul(x[4 * q + c], tw.v[q][c - 1]);
        dft4<1>(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    }
}

// Inverse (includes the 1/256 scale); spectrum order in, x[a] = z_{32a+hl} out.
__device__ __forceinline__ void fft256_inverse(cd (&x)[8], cd *sch, const cd *__restrict__ table,
                                               const LaneTwiddles512 &tw, int hl)
{
    const int hi = hl >> 2, lo = hl & 3;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        dft4<-1>(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
#pragma unroll
        for (int c = 1; c < 4; c++) x[4 * q + c] = cmulc(x[4 * q + c], tw.v[q][c - 1]);
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 4; c++) sch[36 * hi + 4 * (4 * q + lo) + ((c + lo) & 3)] = x[4 * q + c];
    wave_lds_order();
#pragma unroll
    for (int mp = 0; mp < 8; mp++) x[mp] = sch[36 * hi + 4 * mp + ((lo + mp) & 3)];
    wave_lds_order();
    dft8<-1>(x);
    twist_pow<true>(x, tw.l2);
#pragma unroll
    for (int b = 0; b < 8; b++) sch[36 * hi + 4 * b + lo] = x[b];
    wave_lds_order();
#pragma unroll
    for (int m = 0; m < 8; m++) x[m] = sch[36 * m + hl];
    wave_lds_order();
    dft8<-1>(x);
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = cmul(x[a], table[8 + a]);
}

// a[lanes 32..63] <-> b[lanes 0..31], dword by dword (v_permlane32_swap).  With a = this half's
// product for output A and b = its product for output B, a + b afterwards is the complete output A
// in half 0 and the complete output B in half 1.
__device__ __forceinline__ void swap_halves(cd &a, cd &b)
{
    union U { cd c; unsigned w[4]; } x, y;
    x.c = a; y.c = b;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        auto r = __builtin_amdgcn_permlane32_swap(x.w[i], y.w[i], false, false);
        x.w[i] = r[0]; y.w[i] = r[1];
    }
    a = x.c; b = y.c;
}

// Device bootstrapping-key layout for N = 512: cd bsk[n][row 2][part 2][reg 8][hl 32]; half h of the
// wave reads row h (row 0 = the A digits, row 1 = the B digits; L = 1 so row = p).
__host__ __device__ __forceinline__ size_t bsk_index_512(int i, int row, int part, int reg, int hl)
{
    return ((((size_t)i * 2 + row) * 2 + part) * 8 + reg) * 32 + hl;
}

// One external product by bsk[i] of the TRLWE whose polynomial h this half-wave describes through
// coef(j) (j in [0, 512)); returns this half's polynomial of the result as 16 torus words
// (e[a] = coefficient 32a+hl, e[8+a] = coefficient 32a+hl+256).
template <int BGBIT, class F>
__device__ __forceinline__ void external_product_core_512(F coef, uint32_t (&e)[16], const cd *__restrict__ key_i,
                                                          uint32_t offset, cd *sch, const cd *__restrict__ table,
                                                          const LaneTwiddles512 &tw, int h, int hl)
{
    constexpr uint32_t mask = (1u << BGBIT) - 1u;
    constexpr int half = 1 << (BGBIT - 1), shift = 32 - BGBIT;
    TFHE_PRIO(3);                        // phase priorities: see k_blind_rotate_512
    cd ka[8], kb[8];
    const cd *kp = key_i + (size_t)h * 2 * 256 + hl;
#pragma unroll
    for (int k = 0; k < 8; k++) {       // issued first: the L2 latency hides under the forward transform
        ka[k] = kp[k * 32];
        kb[k] = kp[256 + k * 32];
    }
    cd x[8];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const uint32_t d0 = coef(32 * a + hl) + offset, d1 = coef(32 * a + hl + 256) + offset;
        x[a] = cd{(double)((int)((d0 >> shift) & mask) - half), (double)((int)((d1 >> shift) & mask) - half)};
    }
    TFHE_PRIO(1);
    fft256_forward(x, sch, table, tw, hl);
    TFHE_PRIO(0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        cd pa = cmul(x[k], ka[k]), pb = cmul(x[k], kb[k]);
        swap_halves(pa, pb);
        x[k] = pa + pb;
    }
    TFHE_PRIO(2);
    fft256_inverse(x, sch, table, tw, hl);
    TFHE_PRIO(3);
#pragma unroll
    for (int a = 0; a < 8; a++) {
        e[a] = round_to_torus_wide(x[a].re);
        e[8 + a] = round_to_torus_wide(x[a].im);
    }
}

// evaluator.BlindRotateAssign (evaluator.go:110-135) with the gate prep / mod-switch prologue of
// k_blind_rotate.  grid = batch, block = 64 (one wavefront).  Eight of these single-wave workgroups share a CU at full
// launches, two per SIMD and free-running; external_product_core_512 sets an issue priority per phase (key requests +
// decomposition 3, forward transform 1, products 0, inverse transform 2, rounding + update 3: see external_product_core in
// kernels.hpp): Uint2 x 2,048: 3.04 -> 2.71 ms (-11 %), bit-identical (profiles/r03_n_phase_priorities.txt).
template <int BGBIT>
__global__ __launch_bounds__(64, 2) void k_blind_rotate_512(BlindRotateArgs A)
{
    constexpr int N = 512;
    __shared__ cd sc[kScratchSlots512];
    __shared__ uint32_t accL[2][N];
    __shared__ uint16_t abarL[kMaxLweDim];
    __shared__ int btL;
    const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
    const int item = A.first + blockIdx.x;
    if (!gate_item_live(A, item)) return;           // list entries past the device-side count (kernels.hpp)
    const bool bad_op = gate_prep_modswitch(A, item, lane, 64, N, abarL, &btL);
    LaneTwiddles512 tw;
    load_lane_twiddles_512(tw, A.tw, hl);
    __syncthreads();
    uint32_t *acc = accL[h];
    {
        const int bt = btL & (2 * N - 1);
        const uint32_t *tv = A.tv + (size_t)item * A.tv_stride + (size_t)h * N;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = 32 * q + hl;
            const int s = (j - bt) & (2 * N - 1);
            uint32_t v = tv[s & (N - 1)];
            v ^= 0u - (uint32_t)((s >> 9) & 1);
            acc[j] = v;
        }
    }
    __syncthreads();
    cd *sch = sc + h * kHalfScratch;
    for (int i = 0; i < A.nsteps; i++) {
        const int at = __builtin_amdgcn_readfirstlane((int)abarL[i]);
        auto coef = [&](int j) -> uint32_t {
            const int s = (j - at) & (2 * N - 1);
            uint32_t v = acc[s & (N - 1)];
            v ^= 0u - (uint32_t)((s >> 9) & 1);       // "negation" is the bitwise complement
            return v - acc[j];                          // X^at*acc - acc (evaluator.go:93-96,122-126)
        };
        uint32_t e[16];
        external_product_core_512<BGBIT>(coef, e, A.bsk + (size_t)i * 4 * 256, A.offset, sch, A.tw, tw, h, hl);
#pragma unroll
        for (int a = 0; a < 8; a++) {
            lds_add(&acc[32 * a + hl], e[a]);
            lds_add(&acc[32 * a + hl + 256], e[8 + a]);
        }
        wave_lds_order();
    }
    uint32_t *out = A.out + (size_t)item * 2 * N + (size_t)h * N;
#pragma unroll
    for (int q = 0; q < 16; q++) out[32 * q + hl] = acc[32 * q + hl];
    report_bad_op(A, bad_op, lane);
}

// ExternalProductAssign of in[b] with bsk[key_index] (test seam).
template <int BGBIT>
__global__ __launch_bounds__(64) void k_external_product_512(const cd *__restrict__ bsk, const cd *__restrict__ twt,
                                                             int key_index, const uint32_t *__restrict__ in,
                                                             uint32_t *__restrict__ out, uint32_t offset)
{
    constexpr int N = 512;
    __shared__ cd sc[kScratchSlots512];
    const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
    LaneTwiddles512 tw;
    load_lane_twiddles_512(tw, twt, hl);
    const uint32_t *poly = in + (size_t)blockIdx.x * 2 * N + (size_t)h * N;
    auto coef = [&](int j) -> uint32_t { return poly[j]; };
    uint32_t e[16];
    external_product_core_512<BGBIT>(coef, e, bsk + (size_t)key_index * 4 * 256, offset, sc + h * kHalfScratch, twt, tw, h, hl);
    uint32_t *o = out + (size_t)blockIdx.x * 2 * N + (size_t)h * N;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        o[32 * a + hl] = e[a];
        o[32 * a + hl + 256] = e[8 + a];
    }
}

// Reference Fourier layout [n][2][2][512] float64 -> device layout.  One thread per complex.
static __global__ void k_bsk_from_fourier_512(const double *__restrict__ src, cd *__restrict__ dst, int n)
{
    const size_t total = (size_t)n * 2 * 2 * 256;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int hl = idx & 31, reg = (idx >> 5) & 7;
    const size_t polyIdx = idx >> 8;                       // (i*2 + row)*2 + part, same order as the reference
    const double *poly = src + polyIdx * 512;
    const int s = reference_slot_512(reg, hl);
    const int base = 8 * (s >> 2) + (s & 3);
    dst[idx] = cd{poly[base], poly[base + 4]};
}

// Forward transform of `count` torus polynomials, two per wave; dst[poly][reg][hl].
static __global__ __launch_bounds__(64) void k_spectra_512(const uint32_t *__restrict__ src, cd *__restrict__ dst,
                                                           const cd *__restrict__ twt, int count)
{
    __shared__ cd sc[kScratchSlots512];
    const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
    const int polyIdx = 2 * blockIdx.x + h;
    const bool live = polyIdx < count;
    LaneTwiddles512 tw;
    load_lane_twiddles_512(tw, twt, hl);
    const uint32_t *poly = src + (size_t)(live ? polyIdx : 0) * 512;
    cd x[8];
#pragma unroll
    for (int a = 0; a < 8; a++)
        x[a] = cd{(double)(int32_t)poly[32 * a + hl], (double)(int32_t)poly[32 * a + hl + 256]};
    fft256_forward(x, sc + h * kHalfScratch, twt, tw, hl);
    if (!live) return;
#pragma unroll
    for (int k = 0; k < 8; k++) dst[(size_t)polyIdx * 256 + k * 32 + hl] = x[k];
}

// FFT test seams, spectra in the reference FourierPoly layout; two polynomials per wave.
static __global__ __launch_bounds__(64) void k_to_fourier_512(const uint32_t *__restrict__ polys, double *__restrict__ spectra,
                                                              const cd *__restrict__ twt, int count)
{
    __shared__ cd sc[kScratchSlots512];
    const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
    const int polyIdx = 2 * blockIdx.x + h;
    const bool live = polyIdx < count;
    LaneTwiddles512 tw;
    load_lane_twiddles_512(tw, twt, hl);
    const uint32_t *poly = polys + (size_t)(live ? polyIdx : 0) * 512;
    cd x[8];
#pragma unroll
    for (int a = 0; a < 8; a++)
        x[a] = cd{(double)(int32_t)poly[32 * a + hl], (double)(int32_t)poly[32 * a + hl + 256]};
    fft256_forward(x, sc + h * kHalfScratch, twt, tw, hl);
    if (!live) return;
    double *fp = spectra + (size_t)polyIdx * 512;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int s = reference_slot_512(k, hl), base = 8 * (s >> 2) + (s & 3);
        fp[base] = x[k].re;
        fp[base + 4] = x[k].im;
    }
}

static __global__ __launch_bounds__(64) void k_to_poly_512(const double *__restrict__ spectra, uint32_t *__restrict__ polys,
                                                           const cd *__restrict__ twt, int count)
{
    __shared__ cd sc[kScratchSlots512];
    const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
    const int polyIdx = 2 * blockIdx.x + h;
    const bool live = polyIdx < count;
    LaneTwiddles512 tw;
    load_lane_twiddles_512(tw, twt, hl);
    const double *fp = spectra + (size_t)(live ? polyIdx : 0) * 512;
    cd x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int s = reference_slot_512(k, hl), base = 8 * (s >> 2) + (s & 3);
        x[k] = cd{fp[base], fp[base + 4]};
    }
    fft256_inverse(x, sc + h * kHalfScratch, twt, tw, hl);
    if (!live) return;
    uint32_t *poly = polys + (size_t)polyIdx * 512;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        poly[32 * a + hl] = round_to_torus_wide(x[a].re);
        poly[32 * a + hl + 256] = round_to_torus_wide(x[a].im);
    }
}

} // namespace tfhe
