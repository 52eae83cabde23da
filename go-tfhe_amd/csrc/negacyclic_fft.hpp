// negacyclic_fft.hpp -- one-wavefront negacyclic FFT over Z[X]/(X^N+1) for gfx950.
//
// Replaces poly.Evaluator.ToFourierPolyAssign / ToPolyAssignUnsafe of the reference
// (poly/fourier_transform.go:18-21,40-44,178-347).  The reference folds N real
// coefficients into M = N/2 complex points z_j = p_j + i p_{j+M} and evaluates
// Z(X) = sum z_j X^j at the M roots w of X^M = i, i.e. w = zeta^(1+4u), zeta = exp(i pi/N),
// u in Z_M, with a radix-2 scalar loop.  Here ONE 64-lane wavefront owns the whole
// transform with R = M/64 complex points per lane and walks the same root tree in
// radix-8 levels (polynomial remaindering mod X^64 - rho, X^8 - sigma, X - w), each
// level being "pre-twist by a unit root, then an 8-point DFT held in registers":
//
//   N = 1024:           level 1 (a -> m)   exchange   level 2 (b -> m')   exchange   level 3 (c -> m'')
//   point j = 64a+8b+c  reg a, lane 8b+c    --LDS-->   reg b, lane 8m+c    --LDS-->   reg c, lane 8m+m'
//
// and the value left in (reg m'', lane 8m+m') is Z(zeta^(1+4u)), u = m + 8m' + 64m''.
// The spectrum stays in that order: pointwise products do not care, and the
// bootstrapping key is stored in the same order (bsk_index()).  The two lane<->register
// exchanges go through a per-wave LDS scratch with conflict-free padded strides; they
// need no s_barrier because a wave's DS operations execute in issue order.
//
// N = 2048 (Uint5) adds one radix-2 level in registers in front (X^1024 - i splits into
// X^512 -+ exp(i pi/4)) and then runs the two halves as independent 512-point trees.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe {

struct __attribute__((aligned(16))) cd {
    double re, im;
};

__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// a * conj(b)
__device__ __forceinline__ cd cmulc(cd a, cd b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }
__device__ __forceinline__ void cfma(cd &acc, cd a, cd b)
{
    acc.re = fma(a.re, b.re, fma(-a.im, b.im, acc.re));
    acc.im = fma(a.re, b.im, fma(a.im, b.re, acc.im));
}
// a * (S*i)
template <int S> __device__ __forceinline__ cd mul_i(cd a) { return S > 0 ? cd{-a.im, a.re} : cd{a.im, -a.re}; }

// y_k = sum_n x_n exp(S * 2 pi i n k / 8), in place, natural order in and out.
// 52 adds + 8 FMAs: the two 1/sqrt2 rotations of the odd half are not applied to d1, d3
// themselves but folded into the FMAs that consume them (x1,x5 = f0 +- h*s, x3,x7 = f2 +- h*i*t).
// dft8_finish: its second and third stage, from the first stage's sums b0..b3 and differences d0, t1, t2, t3.
template <int S> __device__ __forceinline__ void dft8_finish(cd (&x)[8], const cd b0, const cd b1, const cd b2, const cd b3, const cd d0,
                                                             const cd t1, const cd t2, const cd t3)
{
    constexpr double h = 0.70710678118654752440;
    // u1 = t1 * (1 + S i), u3 = t3 * (-1 + S i)   (sqrt2 * the twiddled values)
    cd u1 = S > 0 ? cd{t1.re - t1.im, t1.im + t1.re} : cd{t1.re + t1.im, t1.im - t1.re};
    cd u3 = S > 0 ? cd{-t3.re - t3.im, t3.re - t3.im} : cd{t3.im - t3.re, -t3.im - t3.re};
    cd d2 = mul_i<S>(t2);
    cd e0 = b0 + b2, e1 = b1 + b3, e2 = b0 - b2, e3 = mul_i<S>(b1 - b3);
    x[0] = e0 + e1; x[4] = e0 - e1; x[2] = e2 + e3; x[6] = e2 - e3;
    cd f0 = d0 + d2, f2 = d0 - d2;
    cd s = u1 + u3, t = mul_i<S>(u1 - u3);
    x[1] = cd{fma(h, s.re, f0.re), fma(h, s.im, f0.im)};
    x[5] = cd{fma(-h, s.re, f0.re), fma(-h, s.im, f0.im)};
    x[3] = cd{fma(h, t.re, f2.re), fma(h, t.im, f2.im)};
    x[7] = cd{fma(-h, t.re, f2.re), fma(-h, t.im, f2.im)};
}
template <int S> __device__ __forceinline__ void dft8(cd (&x)[8])
{
    dft8_finish<S>(x, x[0] + x[4], x[1] + x[5], x[2] + x[6], x[3] + x[7], x[0] - x[4], x[1] - x[5], x[2] - x[6], x[3] - x[7]);
}

// a + b*w as two FMA chains (the product b*w is never formed), and the matching difference a - b*w = 2a - (a + b*w)
__device__ __forceinline__ cd cfma_to(cd a, cd b, cd w)
{
    return {fma(b.re, w.re, fma(-b.im, w.im, a.re)), fma(b.re, w.im, fma(b.im, w.re, a.im))};
}
__device__ __forceinline__ cd twice_minus(cd a, cd s) { return {fma(2.0, a.re, -s.re), fma(2.0, a.im, -s.im)}; }

// Pre-twist by (1, w1, ..., w7), then dft8<S>: a forward level.  The products of the upper four points go straight into the first
// butterfly stage (b = lo + hi*w: four FMAs) and the differences are recovered as 2 lo - b (two FMAs): 20 + 60 instructions
// instead of the 28 + 60 of seven cmul and a dft8.  A recovered difference carries half an ulp of the larger of its two terms, so
// the transform's error stays of the same order: exact at N = 1024, L = 3, outputs are bit-identical.
template <int S>
__device__ __forceinline__ void dft8_pretwist(cd (&x)[8], const cd w1, const cd w2, const cd w3, const cd w4, const cd w5, const cd w6,
                                              const cd w7)
{
    const cd b0 = cfma_to(x[0], x[4], w4), d0 = twice_minus(x[0], b0);
    const cd y1 = cmul(x[1], w1), b1 = cfma_to(y1, x[5], w5), t1 = twice_minus(y1, b1);
    const cd y2 = cmul(x[2], w2), b2 = cfma_to(y2, x[6], w6), t2 = twice_minus(y2, b2);
    const cd y3 = cmul(x[3], w3), b3 = cfma_to(y3, x[7], w7), t3 = twice_minus(y3, b3);
    dft8_finish<S>(x, b0, b1, b2, b3, d0, t1, t2, t3);
}

// A wave's DS operations are executed in issue order, so a wave-private LDS exchange only
// needs the compiler kept from reordering/merging the accesses.
// Waves issuing an LDS exchange run at raised priority so the exchange gets into the (CU-shared) LDS pipe
// early and its latency overlaps the other waves' fp64 work: -4 % blind-rotate time on the batched forward
// exchanges; the same on the inverse / partner exchanges measured +3 %, so only the forward path uses it.
#define TFHE_PRIO(n) __builtin_amdgcn_s_setprio(n)
// (s_setprio has unmodelled side effects: like a volatile asm -- expand_pow below -- it counts as a store to anything when
// LLVM decides whether a wave-uniform load may be a scalar load.  Tying the instruction to a live register with a
// non-volatile asm keeps the loads scalar but lets the instruction float; where both were measured -- k_blind_rotate_2048's
// phase priorities -- the builtin's fixed position was worth more than the scalar loads, 5.22 vs 5.29 ms.)
// LDS-exchange scheduling (r02; A/B on one box, tools/ab_bench.py): the exchanges' DS instructions are spread
// through the arithmetic with sched_group_barrier instead of being issued in bursts -- the waves of these kernels
// spent ~20 % of their cycles stalled on the LDS instruction queue (SQ_WAIT_INST_LDS), a burst of ds_write_b128
// fills it.  k_blind_rotate<3,6,4> x1024: 6.34 -> 6.11 ms; k_blind_rotate_2048 x512: 6.98 -> 6.79 ms;
// k_blind_rotate_quad x128: 3.19 -> 3.10 ms.  The burst forms and the other rejected variants (unpadded scratch,
// spread hand-over stores, static priorities) are recorded in profiles/ and no longer compiled.
constexpr int kPipeValu = 10;       // batched forward transforms: VALU instructions per DS instruction (5 / 8 / 10: 6.15 / 6.13 / 6.11 ms)
constexpr int kPipe1Valu = 3;       // single transforms (N = 2048 blind rotate): VALU instructions between two stores

__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Exchange 1 of a 512-point transform WITHOUT LDS: (reg m; lane 8b + c) <-> (reg b; lane 8m + c) transposes the eight registers against
// lane bits 5, 4, 3 -- v_permlane32_swap for bit 5 (16 instructions for eight complex doubles), v_permlane16_swap for bit 4 (16), and
// for bit 3, where gfx950 has no swap instruction, v_cndmask_b32_dpp row_ror:8 (32): 64 instructions against eight ds_write_b128 +
// eight ds_read_b128.  Eight instructions per store / load pair is too dear wherever the VALU is the busier pipe (the two-wave
// N = 1024 kernel: +0.5 to +4.5 %, profiles/r04_g_radix8_register_exchange.txt; the radix-4 kernels' exchange 1 needs only four per
// pair, kernels_quad.hpp) -- it pays in ONE place: the inverse transform of k_blind_rotate_2048 at two workgroups per CU, the most
// LDS-bound spot of the path (-2.2 % at Uint5 x 512, -2.9 % at x 1,024; the forward transform there: -0.4 %, both: +2 %).
__device__ __forceinline__ void x1_swap32(double &a, double &b)
{
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void x1_swap16(double &a, double &b)
{
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void x1_swap8(cd &a, cd &b)
{
    const int aw[4] = {__double2loint(a.re), __double2hiint(a.re), __double2loint(a.im), __double2hiint(a.im)};
    const int bw[4] = {__double2loint(b.re), __double2hiint(b.re), __double2loint(b.im), __double2hiint(b.im)};
    int na[4], nb[4];
    const unsigned long long clear = 0x00FF00FF00FF00FFull, set = 0xFF00FF00FF00FF00ull;
    asm("s_mov_b64 vcc, %[lo]\n\t"
        "v_cndmask_b32_dpp %[na0], %[b0], %[a0], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[na1], %[b1], %[a1], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[na2], %[b2], %[a2], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[na3], %[b3], %[a3], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %[hi]\n\t"
        "v_cndmask_b32_dpp %[nb0], %[a0], %[b0], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[nb1], %[a1], %[b1], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[nb2], %[a2], %[b2], vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %[nb3], %[a3], %[b3], vcc row_ror:8 row_mask:0xf bank_mask:0xf"
        : [na0] "=&v"(na[0]), [na1] "=&v"(na[1]), [na2] "=&v"(na[2]), [na3] "=&v"(na[3]), [nb0] "=&v"(nb[0]), [nb1] "=&v"(nb[1]),
          [nb2] "=&v"(nb[2]), [nb3] "=&v"(nb[3])
        : [a0] "v"(aw[0]), [a1] "v"(aw[1]), [a2] "v"(aw[2]), [a3] "v"(aw[3]), [b0] "v"(bw[0]), [b1] "v"(bw[1]), [b2] "v"(bw[2]),
          [b3] "v"(bw[3]), [lo] "s"(clear), [hi] "s"(set)
        : "vcc");
    a = cd{__hiloint2double(na[1], na[0]), __hiloint2double(na[3], na[2])};
    b = cd{__hiloint2double(nb[1], nb[0]), __hiloint2double(nb[3], nb[2])};
}
__device__ __forceinline__ void row8_transpose(cd (&x)[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) { x1_swap32(x[i].re, x[i + 4].re); x1_swap32(x[i].im, x[i + 4].im); }
#pragma unroll
    for (int i = 0; i < 8; i++)
        if ((i & 2) == 0) { x1_swap16(x[i].re, x[i + 2].re); x1_swap16(x[i].im, x[i + 2].im); }
#pragma unroll
    for (int i = 0; i < 8; i += 2) x1_swap8(x[i], x[i + 1]);
}

// Per-wave exchange scratch: 8 rows of 72 sixteen-byte slots (64 used + 8 pad).  With this
// stride both exchange patterns are conflict-free for ds_write_b128 (8 contiguous lanes ->
// 8 contiguous slots) and ds_read_b128 (the four 16-lane service groups each touch 16
// distinct slot residues mod 16).
constexpr int kScratchSlots = 8 * 72;
#define SL1W(r) (72 * (r) + lane)
#define SL1R(r) (72 * hi + 8 * (r) + lo)
#define SL2W(r) (72 * hi + 9 * (r) + lo)
#define SL2R(r) (72 * hi + 9 * lo + (r))

// Twiddle table layout (built on the host in long double, tfhe_hip.cpp):
//   [0..7]                 level-1 pre-twists      c1[a]  = zeta^(64 a)          (wave-uniform)
//   [8..15]                inverse level-1 factors conj(c1[a]) / 512
//   [16 + b*64 + lane]     level-2 pre-twists      c2 = zeta^(8 b (1+4m)),        m = lane>>3
//   [16 + 512 + c*64+lane] level-3 pre-twists      c3 = zeta^(c (1+4(m+8m'))),    m' = lane&7
constexpr int kTwLevel2 = 16;
constexpr int kTwLevel3 = 16 + 512;
constexpr int kTwCount1024 = 16 + 1024;

// Per-lane twiddles kept in VGPRs for the life of a kernel: for each of levels 2 and 3 only
// w, w^2, w^4 (w = the b = 1 / c = 1 pre-twist); the other four powers are rebuilt with one
// complex multiply each at every use -- 32 VGPRs traded for 32 fp64 ops per transform, which
// is what lets the key prefetch stay in registers.
struct TwPow {
    cd w1, w2, w4;
};
struct LaneTwiddles {
    TwPow l2, l3;
};

__device__ __forceinline__ void load_lane_twiddles(LaneTwiddles &tw, const cd *__restrict__ table, int lane)
{
    tw.l2.w1 = table[kTwLevel2 + 1 * 64 + lane];
    tw.l2.w2 = table[kTwLevel2 + 2 * 64 + lane];
    tw.l2.w4 = table[kTwLevel2 + 4 * 64 + lane];
    tw.l3.w1 = table[kTwLevel3 + 1 * 64 + lane];
    tw.l3.w2 = table[kTwLevel3 + 2 * 64 + lane];
    tw.l3.w4 = table[kTwLevel3 + 4 * 64 + lane];
}

// x[c] *= w^c (CONJ: conj(w)^c), c = 1..7, from w, w^2, w^4.  The empty asm makes w1 opaque so
// the four derived powers are recomputed here instead of being hoisted out of the CMUX loop
// (which would pin 16 more VGPRs per level and spills; an LDS table of them measured +3 %).
// twist_pow<false> followed by dft8<1>, folded (dft8_pretwist); the same opaque-w1 rebuild of the derived powers
__device__ __forceinline__ void twist_pow_dft8(cd (&x)[8], const TwPow &t)
{
    cd w1 = t.w1;
    asm volatile("" : "+v"(w1.re), "+v"(w1.im));
    const cd w3 = cmul(w1, t.w2), w5 = cmul(w1, t.w4), w6 = cmul(t.w2, t.w4), w7 = cmul(w3, t.w4);
    dft8_pretwist<1>(x, w1, t.w2, w3, t.w4, w5, w6, w7);
}
template <bool CONJ> __device__ __forceinline__ void twist_pow(cd (&x)[8], const TwPow &t)
{
    cd w1 = t.w1;
    asm volatile("" : "+v"(w1.re), "+v"(w1.im));
    const cd w3 = cmul(w1, t.w2), w5 = cmul(w1, t.w4), w6 = cmul(t.w2, t.w4), w7 = cmul(w3, t.w4);
    if (CONJ) {
        x[1] = cmulc(x[1], w1); x[2] = cmulc(x[2], t.w2); x[3] = cmulc(x[3], w3); x[4] = cmulc(x[4], t.w4);
        x[5] = cmulc(x[5], w5); x[6] = cmulc(x[6], w6); x[7] = cmulc(x[7], w7);
    } else {
        x[1] = cmul(x[1], w1); x[2] = cmul(x[2], t.w2); x[3] = cmul(x[3], w3); x[4] = cmul(x[4], t.w4);
        x[5] = cmul(x[5], w5); x[6] = cmul(x[6], w6); x[7] = cmul(x[7], w7);
    }
}

// The same twist with the four derived powers built ONCE by the caller (expand_pow) and shared by the
// transforms of a batch.
struct TwAll {
    cd w3, w5, w6, w7;
};
// The derived powers for a kernel that keeps them for its whole life (built once, outside the CMUX loop): no asm.
// (expand_pow below keeps the rebuild INSIDE a loop with a volatile empty asm; such an asm counts as a store to
// anything for LLVM's "is this uniform load clobbered" test unless it sits in a function with __restrict__ pointer
// parameters, and then turns every wave-uniform twiddle load after it from s_load into global_load + s_waitcnt vmcnt --
// which is what the N = 2048 blind rotate suffered from until round 3.)
__device__ __forceinline__ TwAll expand_pow_once(const TwPow &t)
{
    TwAll a;
    a.w3 = cmul(t.w1, t.w2); a.w5 = cmul(t.w1, t.w4); a.w6 = cmul(t.w2, t.w4);
    a.w7 = cmul(a.w3, t.w4);
    return a;
}
// The rebuild kept inside a loop by a NON-volatile asm whose extra input changes with the loop (dep): see above.
__device__ __forceinline__ TwAll expand_pow(const TwPow &t, int dep)
{
    cd w1 = t.w1;
    asm("" : "+v"(w1.re), "+v"(w1.im) : "s"(dep));
    TwAll a;
    a.w3 = cmul(w1, t.w2); a.w5 = cmul(w1, t.w4); a.w6 = cmul(t.w2, t.w4);
    a.w7 = cmul(a.w3, t.w4);
    return a;
}
__device__ __forceinline__ TwAll expand_pow(const TwPow &t)
{
    cd w1 = t.w1;
    asm volatile("" : "+v"(w1.re), "+v"(w1.im));           // keep the rebuild inside the CMUX loop (see twist_pow)
    TwAll a;
    a.w3 = cmul(w1, t.w2); a.w5 = cmul(w1, t.w4); a.w6 = cmul(t.w2, t.w4);
    a.w7 = cmul(a.w3, t.w4);
    return a;
}
__device__ __forceinline__ void twist_all(cd (&x)[8], const TwPow &t, const TwAll &a)
{
    x[1] = cmul(x[1], t.w1); x[2] = cmul(x[2], t.w2); x[3] = cmul(x[3], a.w3); x[4] = cmul(x[4], t.w4);
    x[5] = cmul(x[5], a.w5); x[6] = cmul(x[6], a.w6); x[7] = cmul(x[7], a.w7);
}
// twist_all followed by dft8<1>, folded (dft8_pretwist)
__device__ __forceinline__ void twist_all_dft8(cd (&x)[8], const TwPow &t, const TwAll &a)
{
    dft8_pretwist<1>(x, t.w1, t.w2, a.w3, t.w4, a.w5, a.w6, a.w7);
}
__device__ __forceinline__ void twist_all_conj(cd (&x)[8], const TwPow &t, const TwAll &a)
{
    x[1] = cmulc(x[1], t.w1); x[2] = cmulc(x[2], t.w2); x[3] = cmulc(x[3], a.w3); x[4] = cmulc(x[4], t.w4);
    x[5] = cmulc(x[5], a.w5); x[6] = cmulc(x[6], a.w6); x[7] = cmulc(x[7], a.w7);
}
// Both levels' derived powers, for kernels with the registers to keep them across a whole CMUX step and share
// them between the step's forward and inverse transforms (the N = 2048 blind rotate: one of each per wave).
struct TwStep {
    TwAll l2, l3;
};

// Index u of the root zeta^(1+4u) held by (reg, lane) after the forward transform.
__host__ __device__ __forceinline__ int spectrum_u_1024(int reg, int lane) { return (lane >> 3) + 8 * (lane & 7) + 64 * reg; }

// Slot of that root in the reference's FourierPoly order: slot s holds
// P(zeta^(1 - 4 bitrev9(s))) (fourier_transform.go:178-247 with the twiddles of
// poly_evaluator.go:114-133), so s = bitrev9(-u mod 512).
__host__ __device__ __forceinline__ int reference_slot_1024(int reg, int lane)
{
    int v = (512 - spectrum_u_1024(reg, lane)) & 511, s = 0;
#pragma unroll
    for (int b = 0; b < 9; b++) s |= ((v >> b) & 1) << (8 - b);
    return s;
}

// Forward transform of 512 complex points; x[a] = z_{64a+lane} in, spectrum order out.
__device__ __forceinline__ void fft512_forward(cd (&x)[8], cd *sc, const cd *__restrict__ table,
                                               const LaneTwiddles &tw, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    dft8_pretwist<1>(x, table[1], table[2], table[3], table[4], table[5], table[6], table[7]);
    // exchange 1: (reg m, lane 8b+c) -> (reg b, lane 8m+c)
#pragma unroll
    for (int m = 0; m < 8; m++) sc[SL1W(m)] = x[m];
    wave_lds_order();
#pragma unroll
    for (int b = 0; b < 8; b++) x[b] = sc[SL1R(b)];
    wave_lds_order();
    twist_pow_dft8(x, tw.l2);
    // exchange 2: (reg m', lane 8m+c) -> (reg c, lane 8m+m')
#pragma unroll
    for (int mp = 0; mp < 8; mp++) sc[SL2W(mp)] = x[mp];
    wave_lds_order();
#pragma unroll
    for (int c = 0; c < 8; c++) x[c] = sc[SL2R(c)];
    wave_lds_order();
    twist_pow_dft8(x, tw.l3);
}

// The same transforms with the derived powers supplied by the caller (TwStep), no rebuild inside.
// Single transforms (one forward, one inverse per wave and step: the N = 2048 blind rotate): every exchange's
// stores are issued one by one under the tail of the arithmetic that produces them (sched_group_barrier)
// instead of as a burst of eight after it.
#define FFT_MIX1(first)                                                           \
    do {                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x2, first, 0);                      \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; k_++) {                        \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x2, kPipe1Valu, 0);             \
        }                                                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                        \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#define FFT_MIX1_BEGIN() __builtin_amdgcn_sched_barrier(0)
// PRIO_IN >= 0: the wave runs the transform's two exchange levels at issue priority PRIO_IN and its last level (and
// whatever follows) at PRIO_OUT -- the phase priorities of k_blind_rotate_2048 (kernels_n2048.hpp).  -1: no s_setprio.
template <int PRIO_IN = -1, int PRIO_OUT = 0>
__device__ __forceinline__ void fft512_forward(cd (&x)[8], cd *sc, const cd *__restrict__ table,
                                               const LaneTwiddles &tw, const TwStep &ts, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    FFT_MIX1_BEGIN();
    if constexpr (PRIO_IN >= 0) TFHE_PRIO(PRIO_IN);
    dft8_pretwist<1>(x, table[1], table[2], table[3], table[4], table[5], table[6], table[7]);
#pragma unroll
    for (int m = 0; m < 8; m++) sc[SL1W(m)] = x[m];
    wave_lds_order();
#pragma unroll
    for (int b = 0; b < 8; b++) x[b] = sc[SL1R(b)];
    wave_lds_order();
    FFT_MIX1(64);
    twist_all_dft8(x, tw.l2, ts.l2);
#pragma unroll
    for (int mp = 0; mp < 8; mp++) sc[SL2W(mp)] = x[mp];
    wave_lds_order();
#pragma unroll
    for (int c = 0; c < 8; c++) x[c] = sc[SL2R(c)];
    wave_lds_order();
    FFT_MIX1(64);
    if constexpr (PRIO_IN >= 0) TFHE_PRIO(PRIO_OUT);
    twist_all_dft8(x, tw.l3, ts.l3);
}
// X1_REG: the second exchange (exchange 1 backwards) in registers (row8_transpose above)
template <int PRIO_IN = -1, int PRIO_OUT = 0, bool X1_REG = false>
__device__ __forceinline__ void fft512_inverse(cd (&x)[8], cd *sc, const cd *__restrict__ table,
                                               const LaneTwiddles &tw, const TwStep &ts, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    FFT_MIX1_BEGIN();
    if constexpr (PRIO_IN >= 0) TFHE_PRIO(PRIO_IN);
    dft8<-1>(x);
    twist_all_conj(x, tw.l3, ts.l3);
#pragma unroll
    for (int c = 0; c < 8; c++) sc[SL2R(c)] = x[c];
    wave_lds_order();
#pragma unroll
    for (int mp = 0; mp < 8; mp++) x[mp] = sc[SL2W(mp)];
    wave_lds_order();
    FFT_MIX1(64);
    dft8<-1>(x);
    twist_all_conj(x, tw.l2, ts.l2);
    if constexpr (X1_REG) {
        row8_transpose(x);
        __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
        for (int b = 0; b < 8; b++) sc[SL1R(b)] = x[b];
        wave_lds_order();
#pragma unroll
        for (int m = 0; m < 8; m++) x[m] = sc[SL1W(m)];
        wave_lds_order();
        FFT_MIX1(64);
    }
    if constexpr (PRIO_IN >= 0) TFHE_PRIO(PRIO_OUT);
    dft8<-1>(x);
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = cmul(x[a], table[8 + a]);
}

// NB independent forward transforms advanced level by level through ONE scratch buffer: the
// wave issues write/read pairs of consecutive transforms back to back (its DS operations
// execute in order, so transform k+1's writes cannot overtake transform k's reads) and only
// waits when the next level's arithmetic needs the data.  This turns 2*NB exposed LDS round
// trips into 2 and gives the scheduler NB-way independent fp64 work per level.
template <int NB>
__device__ __forceinline__ void fft512_forward_batch(cd (&x)[NB][8], cd *sc, const cd *__restrict__ table,
                                                     const LaneTwiddles &tw, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
    for (int t = 0; t < NB; t++) {
        dft8_pretwist<1>(x[t], table[1], table[2], table[3], table[4], table[5], table[6], table[7]);
        TFHE_PRIO(3);
#pragma unroll
        for (int m = 0; m < 8; m++) sc[SL1W(m)] = x[t][m];
        wave_lds_order();
#pragma unroll
        for (int b = 0; b < 8; b++) x[t][b] = sc[SL1R(b)];
        wave_lds_order();
        TFHE_PRIO(0);
    }
    const TwAll a2 = expand_pow(tw.l2);        // once per batch, not per transform: -56 VALU per CMUX step at L = 3
#pragma unroll
    for (int t = 0; t < NB; t++) {
        twist_all_dft8(x[t], tw.l2, a2);
        TFHE_PRIO(3);
#pragma unroll
        for (int mp = 0; mp < 8; mp++) sc[SL2W(mp)] = x[t][mp];
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < 8; c++) x[t][c] = sc[SL2R(c)];
        wave_lds_order();
        TFHE_PRIO(0);
    }
    const TwAll a3 = expand_pow(tw.l3);
#pragma unroll
    for (int t = 0; t < NB; t++) {
        twist_all_dft8(x[t], tw.l3, a3);
    }
}

// Software-pipelined form of fft512_forward_batch: transform t's exchange (8 stores + 8
// loads) is issued INSIDE transform t+1's arithmetic, one DS instruction per few fp64 instructions
// (sched_group_barrier), instead of one burst of 24 + 24 after all three transforms' arithmetic.  Rationale:
// PMC shows the waves of k_blind_rotate spend ~20 % of their cycles stalled on the LDS instruction queue
// (SQ_WAIT_INST_LDS): a burst of ds_write_b128 fills it and blocks the wave, an interleaved stream does not.
template <int NB, class Hook>
__device__ __forceinline__ void fft512_forward_batch_pipe(cd (&x)[NB][8], cd *sc, const cd *__restrict__ table,
                                                          const LaneTwiddles &tw, int lane, Hook before_last_level)
{
    const int hi = lane >> 3, lo = lane & 7;
    auto xchg1 = [&](int t) {
#pragma unroll
        for (int m = 0; m < 8; m++) sc[SL1W(m)] = x[t][m];
        wave_lds_order();
#pragma unroll
        for (int b = 0; b < 8; b++) x[t][b] = sc[SL1R(b)];
        wave_lds_order();
    };
    auto xchg2 = [&](int t) {
#pragma unroll
        for (int mp = 0; mp < 8; mp++) sc[SL2W(mp)] = x[t][mp];
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < 8; c++) x[t][c] = sc[SL2R(c)];
        wave_lds_order();
    };
    auto mix = [&]() {          // the region just written: 1 DS op per kPipeValu VALU ops, stores first
#pragma unroll
        for (int k = 0; k < 8; k++) {
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, kPipeValu, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, kPipeValu, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto level1 = [&](int t) {
        dft8_pretwist<1>(x[t], table[1], table[2], table[3], table[4], table[5], table[6], table[7]);
    };
    __builtin_amdgcn_sched_barrier(0);
    level1(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < NB; t++) { xchg1(t - 1); level1(t); mix(); }
    const TwAll a2 = expand_pow(tw.l2);
    // level 2 of transform t runs over the exchange of the previous one (the last exchange of level 1 first)
#pragma unroll
    for (int t = 0; t < NB; t++) {
        if (t == 0) xchg1(NB - 1); else xchg2(t - 1);
        twist_all_dft8(x[t], tw.l2, a2);
        mix();
    }
    before_last_level();
    const TwAll a3 = expand_pow(tw.l3);
#pragma unroll
    for (int t = 0; t < NB; t++) {
        if (t == 0) { xchg2(NB - 1); }
        twist_all_dft8(x[t], tw.l3, a3);
        if (t == 0) mix();
    }
}

// Inverse transform (includes the 1/512 scale); spectrum order in, x[a] = z_{64a+lane} out.
__device__ __forceinline__ void fft512_inverse(cd (&x)[8], cd *sc, const cd *__restrict__ table,
                                               const LaneTwiddles &tw, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    dft8<-1>(x);
    twist_pow<true>(x, tw.l3);
    // (reg c, lane 8m+m') -> (reg m', lane 8m+c)
#pragma unroll
    for (int c = 0; c < 8; c++) sc[SL2R(c)] = x[c];
    wave_lds_order();
#pragma unroll
    for (int mp = 0; mp < 8; mp++) x[mp] = sc[SL2W(mp)];
    wave_lds_order();
    dft8<-1>(x);
    twist_pow<true>(x, tw.l2);
    // (reg b, lane 8m+c) -> (reg m, lane 8b+c)
#pragma unroll
    for (int b = 0; b < 8; b++) sc[SL1R(b)] = x[b];
    wave_lds_order();
#pragma unroll
    for (int m = 0; m < 8; m++) x[m] = sc[SL1W(m)];
    wave_lds_order();
    dft8<-1>(x);
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = cmul(x[a], table[8 + a]);
}

// fft512_inverse with each exchange's stores issued as soon as their value is final (one ds_write_b128 per twisted
// output, under the remaining twists) instead of in one burst of eight; same arithmetic.
__device__ __forceinline__ void fft512_inverse_pipe(cd (&x)[8], cd *sc, const cd *__restrict__ table,
                                                    const LaneTwiddles &tw, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    auto mix = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x2, 76, 0);            // the 8-point DFT and the first twists
#pragma unroll
        for (int k = 0; k < 8; k++) {
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    __builtin_amdgcn_sched_barrier(0);
    dft8<-1>(x);
    twist_pow<true>(x, tw.l3);
#pragma unroll
    for (int c = 0; c < 8; c++) sc[SL2R(c)] = x[c];
    wave_lds_order();
#pragma unroll
    for (int mp = 0; mp < 8; mp++) x[mp] = sc[SL2W(mp)];
    wave_lds_order();
    mix();
    dft8<-1>(x);
    twist_pow<true>(x, tw.l2);
#pragma unroll
    for (int b = 0; b < 8; b++) sc[SL1R(b)] = x[b];
    wave_lds_order();
#pragma unroll
    for (int m = 0; m < 8; m++) x[m] = sc[SL1W(m)];
    wave_lds_order();
    mix();
    dft8<-1>(x);
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = cmul(x[a], table[8 + a]);
}

// Gadget digits without the subtraction (decomposer.go:60-65: digit_l = ((tmp >> shift_l) & (Bg-1)) - Bg/2).  Flipping the top
// bit of a Bgbit-wide field and reading the field as a two's-complement number IS that subtraction, so with
// tmpx = tmp ^ digit_flip_mask (one XOR per coefficient, all levels at once) every digit is ONE v_bfe_i32:
// -1 VALU instruction per digit, exact on integers (bit-identical outputs).
template <int L, int BGBIT> __host__ __device__ constexpr uint32_t digit_flip_mask()
{
    uint32_t m = 0;
    for (int l = 0; l < L; l++) m |= 1u << (32 - (l + 1) * BGBIT + BGBIT - 1);
    return m;
}
template <int BGBIT> __device__ __forceinline__ int digit_of(uint32_t tmpx, int shift)
{
    return __builtin_amdgcn_sbfe((int)tmpx, shift, BGBIT);
}

// Nearest integer of v, reduced mod 2^32.  Replaces floatModQInPlace + the uint32(int64())
// conversion of the reference (fourier_transform.go:88-125).  Adding 1.5*2^52 leaves the
// rounded integer in the low mantissa bits (two's complement), valid for |v| < 2^51; the
// N=1024 sets are bounded by 2L*N*(Bg/2)*2^31 = 2^48.6.  (Go's math.Round rounds halves
// away from zero, this rounds them to even; a half can only occur once the FFT error
// reaches 0.5, where the reference itself is no longer exact -- SURVEY.md appendix A.)
__device__ __forceinline__ uint32_t round_to_torus_small(double v)
{
    return (uint32_t)__double2loint(v + 6755399441055744.0);
}

// General form for |v| < 2^83 (the Uint sets reach ~2^58).  Adding 1.5*2^84 (ulp 2^32) and taking it
// off again leaves q = v rounded to a multiple of 2^32, exactly; v - q is exact, |v - q| <= 2^31, and
// differs from v by a multiple of 2^32, so the 2^52 trick on it gives round-to-nearest-even(v) mod 2^32
// -- the same value as rint() followed by a floor-based reduction, in 4 additions.  The empty asm keeps
// the compiler from folding (v + M) - M.
__device__ __forceinline__ uint32_t round_to_torus_wide(double v)
{
    double q = v + 29014219670751100192948224.0;      // 1.5 * 2^84
    asm("" : "+v"(q));
    q -= 29014219670751100192948224.0;
    return round_to_torus_small(v - q);
}

} // namespace tfhe
