// kernels_n2048.hpp -- the same path for N = 2048 rings (Uint5: n=1071, L=1, Bgbit=22;
// params/params.go:362-391), used by programmable bootstrapping
// (evaluator/programmable_bootstrap.go:93-115).
//
// M = 1024 complex points per polynomial, 16 per lane.  One radix-2 level in registers splits
// X^1024 - i into X^512 -+ rho (rho = exp(i pi/4)): with j = 64a + lane (a = 0..15) the pair is
// (a, a+8), so no cross-lane traffic; the two halves h = 0,1 are then independent 512-point
// trees run by the same fft512_forward/inverse code with their own twiddle tables.  The value
// left in (half h, reg m'', lane 8m+m') is Z(zeta^(1+4u)), u = h + 2m + 16m' + 128m'',
// zeta = exp(i pi/2048); the reference keeps that root in FourierPoly slot bitrev10(-u mod 1024).
//
// L = 1, so a wave has one forward and one inverse transform per CMUX step and no accumulation
// across levels: the partner's product goes straight to LDS.  Values reach ~2^58 here, beyond
// fp64's 53-bit mantissa: like the reference at this parameter set, results are NOT exact
// integers; the rounding uses the wide form.
#pragma once
#include <type_traits>

#include "kernels.hpp"

namespace tfhe {

constexpr int kScratchSlots2048 = 1152;   // >= 1024 (partner hand-over of a full spectrum), >= 576 (FFT exchanges)

__host__ __device__ __forceinline__ int spectrum_u_2048(int h, int reg, int lane)
{
    return h + 2 * (lane >> 3) + 16 * (lane & 7) + 128 * reg;
}

__host__ __device__ __forceinline__ int reference_slot_2048(int h, int reg, int lane)
{
    int v = (1024 - spectrum_u_2048(h, reg, lane)) & 1023, s = 0;
#pragma unroll
    for (int b = 0; b < 10; b++) s |= ((v >> b) & 1) << (9 - b);
    return s;
}

// bsk[n][2 (p)][1 (L)][2 (part)][16 (h*8 + reg)][64]
__host__ __device__ __forceinline__ size_t bsk_index_2048(int i, int p, int part, int hreg, int lane)
{
    return ((((size_t)(i * 2 + p)) * 2 + part) * 16 + hreg) * 64 + lane;
}

struct LaneTwiddles2048 {
    LaneTwiddles h[2];
};

__device__ __forceinline__ void load_lane_twiddles_2048(LaneTwiddles2048 &tw, const cd *__restrict__ table, int lane)
{
    load_lane_twiddles(tw.h[0], table, lane);
    load_lane_twiddles(tw.h[1], table + kTwCount1024, lane);
}

// x[a] = z_{64a+lane}, a = 0..15 in;  out: x[h*8 + m''] in spectrum order.
__device__ __forceinline__ void fft1024_forward(cd (&x)[16], cd *sc, const cd *__restrict__ table,
                                                const LaneTwiddles2048 &tw, int lane)
{
    constexpr double r = 0.70710678118654752440;
    cd y0[8], y1[8];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const cd v = x[a + 8];
        const cd rv = cd{(v.re - v.im) * r, (v.re + v.im) * r};      // rho * v, rho = (1+i)/sqrt2
        y0[a] = x[a] + rv;                                           // mod X^512 - rho
        y1[a] = x[a] - rv;                                           // mod X^512 + rho
    }
    // the two halves advance level by level through the one scratch buffer (same in-order DS
    // argument as fft512_forward_batch), each with its own twiddle tables
    {
        const int hi = lane >> 3, lo = lane & 7;
        const cd *t0 = table, *t1 = table + kTwCount1024;
        dft8_pretwist<1>(y0, t0[1], t0[2], t0[3], t0[4], t0[5], t0[6], t0[7]);
        TFHE_PRIO(3);
#pragma unroll
        for (int m = 0; m < 8; m++) sc[SL1W(m)] = y0[m];
        wave_lds_order();
#pragma unroll
        for (int b = 0; b < 8; b++) y0[b] = sc[SL1R(b)];
        wave_lds_order();
        TFHE_PRIO(0);
        dft8_pretwist<1>(y1, t1[1], t1[2], t1[3], t1[4], t1[5], t1[6], t1[7]);
        TFHE_PRIO(3);
#pragma unroll
        for (int m = 0; m < 8; m++) sc[SL1W(m)] = y1[m];
        wave_lds_order();
#pragma unroll
        for (int b = 0; b < 8; b++) y1[b] = sc[SL1R(b)];
        wave_lds_order();
        TFHE_PRIO(0);
        twist_pow_dft8(y0, tw.h[0].l2);
        TFHE_PRIO(3);
#pragma unroll
        for (int mp = 0; mp < 8; mp++) sc[SL2W(mp)] = y0[mp];
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < 8; c++) y0[c] = sc[SL2R(c)];
        wave_lds_order();
        TFHE_PRIO(0);
        twist_pow_dft8(y1, tw.h[1].l2);
        TFHE_PRIO(3);
#pragma unroll
        for (int mp = 0; mp < 8; mp++) sc[SL2W(mp)] = y1[mp];
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < 8; c++) y1[c] = sc[SL2R(c)];
        wave_lds_order();
        TFHE_PRIO(0);
        twist_pow_dft8(y0, tw.h[0].l3);
        twist_pow_dft8(y1, tw.h[1].l3);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { x[k] = y0[k]; x[8 + k] = y1[k]; }
}

// Inverse; the per-half tables carry conj(c1)/1024, i.e. the 1/512 of each half and the 1/2 of
// the radix-2 level.
__device__ __forceinline__ void fft1024_inverse(cd (&x)[16], cd *sc, const cd *__restrict__ table,
                                                const LaneTwiddles2048 &tw, int lane)
{
    constexpr double r = 0.70710678118654752440;
    cd y0[8], y1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { y0[k] = x[k]; y1[k] = x[8 + k]; }
    fft512_inverse(y0, sc, table, tw.h[0], lane);
    fft512_inverse(y1, sc, table + kTwCount1024, tw.h[1], lane);
#pragma unroll
    for (int a = 0; a < 8; a++) {
        x[a] = y0[a] + y1[a];
        const cd dlt = y0[a] - y1[a];
        x[a + 8] = cd{(dlt.re + dlt.im) * r, (dlt.im - dlt.re) * r};  // conj(rho) * (y0 - y1)
    }
}

// Polynomial source for the decomposition, N = 2048 flavour of DiffSource.
__device__ __forceinline__ uint32_t diff_coeff_2048(const uint32_t *accL, int at, const uint32_t *plain, int j)
{
    constexpr int N = 2048;
    if (!accL) return plain[j];
    const int s = (j - at) & (2 * N - 1);
    uint32_t v = accL[s & (N - 1)];
    v ^= 0u - (uint32_t)((s >> 11) & 1);
    return v - accL[j];
}

// One external product for L = 1 (evaluator.go:50-81): this wave's half of bsk[i] (x) d.
template <int BGBIT, class Coef>
__device__ __forceinline__ void external_product_core_2048(Coef coef /* j -> coefficient j of the polynomial to decompose */,
                                                           uint32_t (&e)[32], const cd *__restrict__ key_ip,
                                                           cd *sc_mine, const cd *sc_other, const cd *__restrict__ table,
                                                           const LaneTwiddles2048 &tw, uint32_t offset, int p, int lane)
{
    constexpr uint32_t mask = (1u << BGBIT) - 1u;
    constexpr int half = 1 << (BGBIT - 1);
    constexpr int shift = 32 - BGBIT;
    cd x[16];
#pragma unroll
    for (int a = 0; a < 16; a++) {
        const uint32_t d0 = coef(64 * a + lane) + offset;
        const uint32_t d1 = coef(64 * a + lane + 1024) + offset;
        x[a] = cd{(double)((int)((d0 >> shift) & mask) - half), (double)((int)((d1 >> shift) & mask) - half)};
    }
    fft1024_forward(x, sc_mine, table, tw, lane);
    const cd *kKeep = key_ip + (size_t)(p ? 1 : 0) * 1024 + lane;   // wave p keeps output p
    const cd *kSend = key_ip + (size_t)(p ? 0 : 1) * 1024 + lane;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        sc_mine[k * 64 + lane] = cmul(x[k], kSend[k * 64]);        // partner's share straight to LDS
        x[k] = cmul(x[k], kKeep[k * 64]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = x[k] + sc_other[k * 64 + lane];
    __syncthreads();
    fft1024_inverse(x, sc_mine, table, tw, lane);
#pragma unroll
    for (int a = 0; a < 16; a++) {
        e[a] = round_to_torus_wide(x[a].re);
        e[a + 16] = round_to_torus_wide(x[a].im);
    }
}

// Extended lookup tables (see ExtendedArgs below): rotation amount s and source component src for target component k of
// Y^a * acc, a = ext*q + r: component k comes from component (k - r) mod ext rotated by X^(q + [k < r]).
__device__ __forceinline__ void ext_rotation(int a, int ext, int k, int &src, int &s)
{
    const int q = a / ext, r = a - q * ext;
    src = k - r; if (src < 0) src += ext;
    s = q + (k < r ? 1 : 0);
}

// Blind rotate for N = 2048 with FOUR waves per bootstrap (one workgroup): wave (p, h) owns half h of the root tree
// (X^512 = +-rho) of accumulator polynomial p.  A batch of 512 PBS puts two workgroups on each CU = two waves on every
// SIMD (fp64 issue ~5.5 instead of ~8 cycles per instruction).  Per CMUX step and wave, FOUR s_barriers:
//
//   extract   the digits of the wave's OWN 16 coefficients of X^a~ * acc - acc (points a in [4h, 4h+4) and a + 8, re and
//             im): rotated operand from the polynomial's accumulator in LDS, own operand from registers; fold them
//             to both half-trees (lo +- rho*hi), keep this half-tree's four inputs, hand the other four to the sibling
//   barrier 1
//   forward   512-point transform of the half-tree, products with the 8 + 8 key slices: the partner polynomial's
//             share to LDS, its own kept
//   barrier 2
//   gather + inverse   own + partner's share, 512-point inverse transform THROUGH THE PARTNER'S SCRATCH (what that held
//             -- the partner's products for this wave -- was just read by this wave itself, so nothing has to be
//             waited for: this removed a barrier); the four results the sibling's points need go to LDS
//   barrier 3
//   update    undo the radix-2 level for the own points, acc += round(.) in registers and in the LDS accumulator
//   barrier 4
//
// LDS per bootstrap: four exchange scratches (36 KB: FFT exchanges, product hand-over, and -- in the windows where
// their owner does not use them -- the digit and half-swap hand-overs), the accumulator (16 KB), the mod-switched
// mask (2.5 KB): 55.8 KB, two workgroups per CU (the registers, 218, allow no third).
// Measured (tools/ab_bench.py, Uint5 x 512, interleaved on one box; profiles/r03_a_uint5_steps.txt): 6.66 ms at the
// start of round 3 -> 6.50 (own points in registers, half-swap of 4 slots instead of 8, scalar twiddle loads)
// -> 6.43 (inverse through the partner's scratch: four barriers) -> 6.02 (one bootstrap per workgroup again: with four
// barriers two unsynchronised workgroups per CU beat one eight-wave workgroup) -> 5.94 (hand-overs inside the scratches,
// twiddle powers built once) -> 5.85 (the step loop instantiated once per half-tree h: the two h-dependent hand-over
// patterns are static, 32 v_cndmask per step and a block boundary gone; 5.96 -> 5.85 interleaved on a later, slower
// box -- FMA contraction moves again, i.e. the bits change inside the tolerance regime).  Tried and dropped: a signed table {acc, ~acc} for the rotated reads (-100 VALU per
// step but 16 more LDS stores: 6.18 ms -- the kernel is bound by LDS traffic before it is bound by issue slots); key
// slices requested at the top of the step at two workgroups per CU (6.85 ms; it is what the <= 256 launches use); a
// forced half-step offset between the two workgroups of a CU (6.11-6.15 ms).
// KEYS_FIRST: the step's 16 key slices are requested at its top and held in 64 VGPRs (launches of at most one workgroup
// per CU, where nothing else covers the L2 latency: 4.76 -> 4.34 ms at 256); with two workgroups per CU they are
// requested where they are used, under the last level of the forward transform (6.10 vs 6.15 ms at 512).
//
// EXT = 2: the same kernel over an EXTENDED lookup table of 2N entries (polyExtendFactor 2: the Uint6 set, params.go:396-403;
// the algebra is at ExtendedArgs below).  The table is two ring elements, the workgroup two four-wave groups, group c
// holding accumulator component c.  Every word is mod-switched to [0, 4N); a step with a = 2q + r updates
//     acc_c <- acc_c + bsk[i] (x) (X^(q + [c < r]) acc_((c - r) mod 2) - acc_c):
// the only difference to EXT = 1 is WHICH component's signed table the rotated operand is read from, and by how much.
// All eight waves share the four barriers, so every rotated read of a step precedes every update of that step.
// 141 KB of LDS: one bootstrap per CU, two waves per SIMD.  A.tv = lut [2][2][N] (+ tv_stride per item), A.in1 unused.
template <int BGBIT, bool KEYS_FIRST, int EXT = 1>
__global__ __launch_bounds__(256 * EXT) void k_blind_rotate_2048(BlindRotateArgs A)
{
    constexpr int N = 2048;
    constexpr double r = 0.70710678118654752440;
    __shared__ cd scAll[EXT][4][kScratchSlots];
    __shared__ uint32_t accL[EXT][2][N];          // the accumulator polynomials (per component): the operand of the ROTATED reads
    __shared__ uint16_t abarL[kMaxLweDim];
    __shared__ int btL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wAll = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int w = wAll & 3, comp = EXT > 1 ? wAll >> 2 : 0;
    const int p = w >> 1, h = w & 1;
    cd (&sc)[4][kScratchSlots] = scAll[comp];
    const int item = A.first + blockIdx.x;
    if (!gate_item_live(A, item)) return;           // list entries past the device-side count (kernels.hpp)
    const int n = A.n;
    bool bad_op = false;
    if constexpr (EXT == 1) {
        bad_op = gate_prep_modswitch(A, item, tid, 256, N, abarL, &btL);
    } else {
        // round(ct * 2*EXT*N / 2^32) for every word; the body holds 2*EXT*N - that (evaluator.go:116,122 on the big ring)
        const uint32_t *ct = A.in0 + (size_t)item * (n + 1);
        constexpr unsigned long long big2 = 2ull * EXT * N;
        for (int x = tid; x <= n; x += 256 * EXT) {
            unsigned int a = (unsigned int)(((unsigned long long)ct[x] * big2 + (1ull << 31)) >> 32);
            if (a >= big2) a -= (unsigned int)big2;
            if (x == n) btL = (int)((big2 - a) % big2);
            else abarL[x] = (uint16_t)a;
        }
    }
    LaneTwiddles tw;
    const cd *table = A.tw + (size_t)h * kTwCount1024;
    load_lane_twiddles(tw, table, lane);
    // Two workgroups per CU: the loop carries s_setprio builtins, which de-scalarise in-loop wave-uniform loads -- the 15 level-1
    // twiddles of a step came back as vector loads with a wait each.  They live in scalar registers for the whole kernel instead
    // (loaded once, here; 60 SGPRs): Uint5 x 512 5.125 -> 5.056 ms.  (Not the extended-table form: there the same hoist spills,
    // 5.5 -> 8.3 ms; and the one-workgroup-per-CU form has no s_setprio and scalar loads anyway.)
    cd Tu[16];
    if constexpr (!KEYS_FIRST && EXT == 1) {
#pragma unroll
        for (int k = 0; k < 16; k++) Tu[k] = table[k];
        table = Tu;
    }
    __syncthreads();
    uint32_t *T = accL[comp][p];
    // Wave h owns the digit points a in [4h, 4h+4) and a + 8 of its polynomial, i.e. coefficients
    // j0 = 256h + 64q + lane (q < 4) and j0 + 1024 (re, im of point a), j0 + 512 and j0 + 1536 (point a + 8): it
    // extracts their digits AND applies their updates, so it keeps them in registers next to the table.
    uint32_t own[4][4];
    auto own_j = [&](int q, int k) { return 256 * h + 64 * q + lane + 512 * (k >> 1) + 1024 * (k & 1); };
    {
        // acc = X^b~ * testvec (evaluator.go:116-118, buffer_methods.go:133-164); EXT > 1: component comp of Y^b~ * LUT
        int bt = btL & (2 * N - 1), src = 0;
        if constexpr (EXT > 1) ext_rotation(btL, EXT, comp, src, bt);
        const uint32_t *tv = A.tv + (size_t)item * A.tv_stride + ((size_t)src * 2 + p) * N;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = own_j(q, k), s = (j - bt) & (2 * N - 1);
                uint32_t v = tv[s & (N - 1)];
                v ^= 0u - (uint32_t)((s >> 11) & 1);      // "negation" is the bitwise complement
                own[q][k] = v;
                T[j] = v;
            }
    }
    __syncthreads();
    constexpr uint32_t mask = (1u << BGBIT) - 1u;
    constexpr int half = 1 << (BGBIT - 1);
    constexpr int shift = 32 - BGBIT;
    const cd *key = A.bsk + (size_t)p * 2 * 1024 + (size_t)h * 8 * 64 + lane;
    constexpr size_t kStep = (size_t)2 * 2 * 1024;
    const int wpart = ((1 - p) << 1) | h;               // same half of the other polynomial
    const int wsib = w ^ 1;                             // other half of the same polynomial
    const double sr = h ? -r : r;
    // hand-over windows inside the exchange scratches (see the header): digits for the sibling go into the SIBLING's
    // scratch (idle until its owner's forward transform, which reads them first); half-swap values into the scratch
    // this wave's inverse transform has just finished with (the partner's), where the sibling finds them
    cd *dsend = sc[wsib], *drecv = sc[w];
    cd *ssend = sc[wpart], *srecv = sc[wpart ^ 1];
    // both levels' derived twiddle powers, kept for the whole kernel: they are live through every step's two transforms
    // anyway, so rebuilding them per step bought no registers at the peak
    const TwStep ts{expand_pow_once(tw.l2), expand_pow_once(tw.l3)};
    PhaseClock clk;
    clk.start();
    // Issue priorities by phase (s_setprio), for the launches with two waves per SIMD (two free-running workgroups per CU,
    // or the eight waves of the extended-table form): extract 3, forward
    // exchanges 1, last forward level + products + gather 0, inverse exchanges 2, last inverse level + half swap 0, update 3.
    // The two waves of a SIMD belong to different workgroups and are rarely in the same phase; the arbiter then favours
    // the wave whose phase ends in a barrier three other waves wait at, and keeps the two workgroups out of step (one in
    // its LDS-heavy exchanges while the other issues fp64).  Uint5 x 512: 5.85 -> 5.21 ms (profiles/r03_n_phase_priorities.txt;
    // equal priorities for all phases 5.65, none 5.85).  At one workgroup per CU there is nothing to arbitrate and the
    // s_setprio instructions only cut the scheduling regions (4.19 -> 4.42 ms at 256): KEYS_FIRST instances run without.
    // With the builtins in the loop its 15 wave-uniform level-1 twiddle loads per step are vector loads again (see
    // TFHE_PRIO in negacyclic_fft.hpp; tests/test_codegen.py carries the counts): measured, that costs less than it gains.
    // Extended tables (EXT = 2, one eight-wave workgroup per CU): 6.09 -> 5.87 ms at 64.
    constexpr bool kPhasePrio = !KEYS_FIRST;
    auto steps = [&](auto half_tag) {
    constexpr int H = decltype(half_tag)::value;        // = h, as a constant: the two hand-over patterns below are
                                                        // static per instance instead of 32 v_cndmask per step
    for (int i = 0; i < A.nsteps; i++) {
        int at = __builtin_amdgcn_readfirstlane((int)abarL[i]);
        if constexpr (kPhasePrio) TFHE_PRIO(3);          // extract
        const uint32_t *Trot = T;                       // table the rotated operand is read from
        if constexpr (EXT > 1) {
            int src;
            ext_rotation(at, EXT, comp, src, at);
            Trot = accL[src][p];
        }
        const cd *kp = key + (size_t)i * kStep;
        const cd *kKeep = kp + (size_t)(p ? 1 : 0) * 1024;
        const cd *kSend = kp + (size_t)(p ? 0 : 1) * 1024;
        cd kk[8], ks[8];
        if constexpr (KEYS_FIRST) {
#pragma unroll
            for (int k = 0; k < 8; k++) { kk[k] = kKeep[k * 64]; ks[k] = kSend[k * 64]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        cd keep[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int dg[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {            // k = 0: re of point a, 1: im of a, 2: re of a + 8, 3: im of a + 8
                const int sx = (own_j(q, k) - at) & (2 * N - 1);
                const uint32_t v = Trot[sx & (N - 1)] ^ (0u - (uint32_t)((sx >> 11) & 1));     // "negation" = complement
                const uint32_t d = v - own[q][k] + A.offset;         // X^at*acc - acc (evaluator.go:93-96), + offset
                dg[k] = (int)((d >> shift) & mask) - half;           // decomposer.go:60-65
            }
            // rho*(a + ib) = ((a - b) + i(a + b))/sqrt2, a -+ b formed on the integer digits (exact: |digit| < 2^21);
            // sr = +-1/sqrt2 by half-tree: y_h[a] = lo + (+-rho)*hi is kept, lo - (+-rho)*hi goes to the other wave
            // (written as FMAs with a uniform sign: a select between the two sums ends up in scratch memory)
            const double lo_re = (double)dg[0], lo_im = (double)dg[1];
            const double t_re = (double)(dg[2] - dg[3]), t_im = (double)(dg[2] + dg[3]);
            keep[q] = cd{fma(sr, t_re, lo_re), fma(sr, t_im, lo_im)};
            dsend[q * 64 + lane] = cd{fma(-sr, t_re, lo_re), fma(-sr, t_im, lo_im)};
        }
        clk.mark(0);
        __syncthreads();
        clk.mark(1);
        cd y[8];
        if constexpr (H == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) { y[q] = keep[q]; y[4 + q] = drecv[q * 64 + lane]; }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) { y[4 + q] = keep[q]; y[q] = drecv[q * 64 + lane]; }
        }
        fft512_forward<kPhasePrio ? 1 : -1, 0>(y, sc[w], table, tw, ts, lane);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if constexpr (!KEYS_FIRST) { kk[k] = kKeep[k * 64]; ks[k] = kSend[k * 64]; }
            sc[w][k * 64 + lane] = cmul(y[k], ks[k]);
            y[k] = cmul(y[k], kk[k]);
        }
        clk.mark(2);
        __syncthreads();
        clk.mark(3);
#pragma unroll
        for (int k = 0; k < 8; k++) y[k] = y[k] + sc[wpart][k * 64 + lane];
        clk.mark(4);
        // two workgroups per CU (kPhasePrio): the inverse's second exchange runs in registers, not through LDS (negacyclic_fft.hpp,
        // row8_transpose: -2.2 % at Uint5 x 512, -6.9 % for the extended-table form at Uint6 x 64; at one four-wave workgroup per CU it costs 0.9 %)
        fft512_inverse<kPhasePrio ? 2 : -1, 0, kPhasePrio>(y, sc[wpart], table, tw, ts, lane);  // table carries conj(c1)/1024
        // undo the radix-2 level, x[a] = y0[a] + y1[a], x[a+8] = conj(rho)(y0[a] - y1[a]), for the wave's OWN points
        // a = 4h + q: it sends the sibling's four values and receives its own four -- half the traffic of exchanging
        // all eight, and both waves do the same work
        cd mine[4];                         // (uniform branches, not selects: those end up in scratch memory)
        if constexpr (H == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) { ssend[q * 64 + lane] = y[4 + q]; mine[q] = y[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) { ssend[q * 64 + lane] = y[q]; mine[q] = y[4 + q]; }
        }
        clk.mark(6);
        __syncthreads();
        clk.mark(7);
        if constexpr (kPhasePrio) TFHE_PRIO(3);          // update
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const cd o = srecv[q * 64 + lane], m = mine[q];
            const cd sm = m + o, dl = m - o;                               // m - o = (-1)^h (y0 - y1)
            const double z[4] = {sm.re, sm.im, (dl.re + dl.im) * sr, (dl.im - dl.re) * sr};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                own[q][k] += round_to_torus_wide(z[k]);                    // acc += e (evaluator.go:102-105)
                T[own_j(q, k)] = own[q][k];
            }
        }
        clk.mark(8);
        __syncthreads();
        clk.mark(9);
    }
    };
    if (h == 0) steps(std::integral_constant<int, 0>{});
    else steps(std::integral_constant<int, 1>{});
#ifdef PHASE_TRACE
    clk.store(A.out + (size_t)item * 2 * N, w, lane);
    return;
#endif
    if (comp == 0) {                                    // sample extraction reads coefficient 0 of component 0 only
        uint32_t *out = A.out + (size_t)item * 2 * N + (size_t)p * N;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int k = 0; k < 4; k++) out[own_j(q, k)] = own[q][k];
    }
    report_bad_op(A, bad_op, tid);
}

template <int BGBIT>
__global__ __launch_bounds__(128) void k_external_product_2048(const cd *bsk, const cd *twt, int key_index,
                                                                   const uint32_t *in, uint32_t *out, uint32_t offset)
{
    constexpr int N = 2048;
    __shared__ cd sc[2][kScratchSlots2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
    LaneTwiddles2048 tw;
    load_lane_twiddles_2048(tw, twt, lane);
    const uint32_t *src = in + (size_t)blockIdx.x * 2 * N + (size_t)p * N;
    uint32_t e[32];
    const cd *key = bsk + ((size_t)key_index * 2 + p) * 2 * 1024;
    external_product_core_2048<BGBIT>([&](int j) { return src[j]; }, e, key, sc[p], sc[p ^ 1], twt, tw, offset, p, lane);
    uint32_t *dst = out + (size_t)blockIdx.x * 2 * N + (size_t)p * N;
#pragma unroll
    for (int q = 0; q < 32; q++) dst[64 * q + lane] = e[q];
}

// ------------------------------------------------------------------------------------
// Extended lookup tables (LookUpTableSize = ext * N > N; params.go:399-402,440-443,481-484: Uint6/7/8 carry the
// parameters, the reference leaves the algorithm out -- params/UINT_STATUS.md:12-30).  The table is a polynomial
// P(Y) of degree ext*N over Y^(ext*N) = -1, kept as ext ring elements p_k(X), X = Y^ext:
// P(Y) = sum_k Y^k p_k(Y^ext).  Multiplying by Y^a, a = ext*q + r, sends component k to component (k + r) mod ext
// rotated by X^q (X^(q+1) when k + r wrapped), so one CMUX step of the blind rotation is ext external products:
//     acc_k <- acc_k + bsk[i] (x) ( X^(q + [k < r]) * acc_((k - r) mod ext) - acc_k ),   a = modswitch(ct[i]) in [0, 2 ext N).
// ext = 1 is exactly evaluator.BlindRotateAssign (evaluator.go:110-135).  One launch per step over B * ext
// (item, component) pairs, accumulators double-buffered in global memory [ext][B][2][N]: a functional path for the
// experimental sets (launch-bound: ~n launches per batch), not a tuned one.
// ------------------------------------------------------------------------------------
struct ExtendedArgs {
    const cd *bsk, *tw;
    const uint32_t *in;        // [B][n+1] LWE samples
    const uint32_t *lut;       // [ext][2][N] (shared) or [B][ext][2][N]
    long lut_stride;           // 0 or ext*2*N
    uint32_t *amod;            // [B][n+1]: every word mod-switched to [0, 2*ext*N); the body holds 2*ext*N - modswitch(b)
    const uint32_t *acc_in;    // [ext][B][2][N]
    uint32_t *acc_out;
    int n, ext, B, step;
    uint32_t offset;
};

__device__ __forceinline__ uint32_t rot_coeff_2048(const uint32_t *__restrict__ poly, int s, int j)
{
    const int t = (j - s) & 4095;                       // X^s * poly, X^2048 = -1; "negation" is the complement
    return poly[t & 2047] ^ (0u - (uint32_t)((t >> 11) & 1));
}

// mod-switch of every word to the big ring and the initial accumulators acc_k = component k of Y^(b~) * LUT.
static __global__ void k_ext_init_2048(ExtendedArgs A)
{
    constexpr int N = 2048;
    const int item = blockIdx.x, n = A.n, ext = A.ext;
    const uint32_t *ct = A.in + (size_t)item * (n + 1);
    const unsigned long long big2 = 2ull * ext * N;
    __shared__ int bt;
    for (int x = threadIdx.x; x <= n; x += blockDim.x) {
        // round(ct * 2*ext*N / 2^32): for ext a power of two this is (ct + rnd) >> sh of evaluator.go:116,122
        unsigned int a = (unsigned int)(((unsigned long long)ct[x] * big2 + (1ull << 31)) >> 32);
        if (a >= big2) a -= (unsigned int)big2;
        if (x == n) { a = (unsigned int)((big2 - a) % big2); bt = (int)a; }
        A.amod[(size_t)item * (n + 1) + x] = a;
    }
    __syncthreads();
    const uint32_t *lut = A.lut + (size_t)item * A.lut_stride;
    for (int k = 0; k < ext; k++) {
        int src, s;
        ext_rotation(bt, ext, k, src, s);
        for (int part = 0; part < 2; part++) {
            uint32_t *dst = A.acc_out + (((size_t)k * A.B + item) * 2 + part) * N;
            const uint32_t *sp = lut + ((size_t)src * 2 + part) * N;
            for (int j = threadIdx.x; j < N; j += blockDim.x) dst[j] = rot_coeff_2048(sp, s, j);
        }
    }
}

template <int BGBIT>
__global__ __launch_bounds__(128) void k_cmux_ext_2048(ExtendedArgs A)
{
    constexpr int N = 2048;
    __shared__ cd sc[2][kScratchSlots2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ext = A.ext, item = blockIdx.x / ext, k = blockIdx.x - item * ext;
    LaneTwiddles2048 tw;
    load_lane_twiddles_2048(tw, A.tw, lane);
    int src, s;
    ext_rotation((int)A.amod[(size_t)item * (A.n + 1) + A.step], ext, k, src, s);
    const uint32_t *cur = A.acc_in + (((size_t)k * A.B + item) * 2 + p) * N;
    const uint32_t *rot = A.acc_in + (((size_t)src * A.B + item) * 2 + p) * N;
    uint32_t e[32];
    const cd *key = A.bsk + ((size_t)A.step * 2 + p) * 2 * 1024;
    external_product_core_2048<BGBIT>([&](int j) { return rot_coeff_2048(rot, s, j) - cur[j]; }, e, key, sc[p], sc[p ^ 1],
                                      A.tw, tw, A.offset, p, lane);
    uint32_t *dst = A.acc_out + (((size_t)k * A.B + item) * 2 + p) * N;
#pragma unroll
    for (int q = 0; q < 32; q++) dst[64 * q + lane] = cur[64 * q + lane] + e[q];
}

// Reference Fourier layout [n][2][2][2048] float64 -> device layout (L = 1).
static __global__ void k_bsk_from_fourier_2048(const double *__restrict__ src, cd *__restrict__ dst, int n)
{
    const size_t total = (size_t)n * 2 * 2 * 1024;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, hreg = (idx >> 6) & 15, part = (idx >> 10) & 1;
    const size_t ip = idx >> 11;                 // i*2 + p ; reference row r = p (L = 1)
    const double *poly = src + (ip * 2 + part) * 2048;
    const int s = reference_slot_2048(hreg >> 3, hreg & 7, lane);
    const int base = 8 * (s >> 2) + (s & 3);
    dst[idx] = cd{poly[base], poly[base + 4]};
}

static __global__ __launch_bounds__(64) void k_bsk_from_torus_2048(const uint32_t *__restrict__ src, cd *__restrict__ dst,
                                                             const cd *__restrict__ twt)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    const int polyIdx = blockIdx.x;              // ((i*2 + r)*2 + part), r = p
    LaneTwiddles2048 tw;
    load_lane_twiddles_2048(tw, twt, lane);
    const uint32_t *poly = src + (size_t)polyIdx * 2048;
    cd x[16];
#pragma unroll
    for (int a = 0; a < 16; a++)
        x[a] = cd{(double)(int32_t)poly[64 * a + lane], (double)(int32_t)poly[64 * a + lane + 1024]};
    fft1024_forward(x, sc, twt, tw, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) dst[((size_t)polyIdx * 16 + k) * 64 + lane] = x[k];
}

static __global__ __launch_bounds__(64) void k_to_fourier_2048(const uint32_t *__restrict__ polys, double *__restrict__ spectra,
                                                         const cd *__restrict__ twt)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    LaneTwiddles2048 tw;
    load_lane_twiddles_2048(tw, twt, lane);
    const uint32_t *poly = polys + (size_t)blockIdx.x * 2048;
    double *fp = spectra + (size_t)blockIdx.x * 2048;
    cd x[16];
#pragma unroll
    for (int a = 0; a < 16; a++)
        x[a] = cd{(double)(int32_t)poly[64 * a + lane], (double)(int32_t)poly[64 * a + lane + 1024]};
    fft1024_forward(x, sc, twt, tw, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int s = reference_slot_2048(k >> 3, k & 7, lane), base = 8 * (s >> 2) + (s & 3);
        fp[base] = x[k].re;
        fp[base + 4] = x[k].im;
    }
}

static __global__ __launch_bounds__(64) void k_to_poly_2048(const double *__restrict__ spectra, uint32_t *__restrict__ polys,
                                                      const cd *__restrict__ twt)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    LaneTwiddles2048 tw;
    load_lane_twiddles_2048(tw, twt, lane);
    const double *fp = spectra + (size_t)blockIdx.x * 2048;
    uint32_t *poly = polys + (size_t)blockIdx.x * 2048;
    cd x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int s = reference_slot_2048(k >> 3, k & 7, lane), base = 8 * (s >> 2) + (s & 3);
        x[k] = cd{fp[base], fp[base + 4]};
    }
    fft1024_inverse(x, sc, twt, tw, lane);
#pragma unroll
    for (int a = 0; a < 16; a++) {
        poly[64 * a + lane] = round_to_torus_wide(x[a].re);
        poly[64 * a + lane + 1024] = round_to_torus_wide(x[a].im);
    }
}

} // namespace tfhe
