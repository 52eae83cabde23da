// kernels_quad.hpp -- small-batch blind rotate for the N = 1024 rings: FOUR wavefronts per bootstrap.
//
// k_blind_rotate (kernels.hpp) gives a bootstrap two waves; a launch of up to one workgroup per CU then
// leaves two of the four SIMDs of every CU idle and each wave issues ~1,550 VALU instructions per CMUX
// step at the ~8 cycles a lone wave needs per fp64 instruction: 5.0 us per step, 3.5 ms per blind rotate
// whatever the batch (1..256).  Here wave (p, h) owns half h of the root tree of accumulator polynomial p:
//
//   X^512 - i = (X^256 - rho)(X^256 + rho), rho = exp(i pi/4):  y_h[j] = z_j + (-1)^h rho z_{j+256}, j < 256,
//
// and the two halves are independent 256-point transforms, run with FOUR points per lane in radix-4 levels
// (polynomial remaindering mod X^64 - sigma, X^16 - tau, X^4 - upsilon, X - w: pre-twist by a unit root, then a
// 4-point DFT in registers) and three lane<->register exchanges through a wave-private, unpadded,
// conflict-free 4 KiB LDS scratch:
//
//   point j = 64a+16b+4c+d   level 1 (a -> m)  xchg   level 2 (b -> m')  xchg   level 3 (c -> m'')  xchg   level 4 (d -> m''')
//   reg a, lane 16b+4c+d     -->  reg b, lane 16m+4c+d  -->  reg c, lane 16m+4m'+d  -->  reg d, lane 16m+4m'+m''
//
// leaving Z(zeta^(1+4u)), u = h + 2m + 8m' + 32m'' + 128m''', in (reg m''', lane 16m+4m'+m''); the
// bootstrapping key is kept in that order as well (bskq_index).  Per CMUX step a wave runs L forward and one
// inverse 256-point transform -- about 40 % of the instructions of a k_blind_rotate wave -- and meets the
// others at TWO barriers: the partner-polynomial sum with wave (1-p, h) and the half swap with wave (p, 1-h)
// that undoes the radix-2 level.  Every wave keeps a PRIVATE copy of its polynomial's accumulator (both
// siblings compute the full update from the two half results; the additions are symmetric, so the copies stay
// bit-identical), which removes the third barrier an accumulator shared by the siblings would need.
// (evaluator.go:50-135; fourier_transform.go:178-347; decomposer.go:55-66)
#pragma once

#include "kernels.hpp"

namespace tfhe {

// Twiddle table, one block of kTwQuadHalf entries per half h (host: make_twiddles_quad):
//   [0..3]   level-1 pre-twists  zeta^(64 a (1+4h))                    (wave-uniform)
//   [4..7]   their conjugates / 512 (inverse level 1: 1/256 of the transform, 1/2 of the radix-2 level)
//   [8 + ((lvl*3 + a-1)*64 + lane)], lvl = 0,1,2, a = 1,2,3:
//            level-2 zeta^(16 a (1+4(h+2m))), level-3 zeta^(4 a (1+4(h+2m+8m'))), level-4 zeta^(a (1+4(h+2m+8m'+32m'')))
//            with m = lane>>4, m' = (lane>>2)&3, m'' = lane&3.
constexpr int kTwQuadHalf = 8 + 9 * 64;

struct QuadTwiddles {
    cd w[3][3];      // [level 2..4][a-1]
};

__device__ __forceinline__ void load_quad_twiddles(QuadTwiddles &tw, const cd *__restrict__ table /* half h */, int lane)
{
#pragma unroll
    for (int lvl = 0; lvl < 3; lvl++)
#pragma unroll
        for (int a = 0; a < 3; a++) tw.w[lvl][a] = table[8 + (lvl * 3 + a) * 64 + lane];
}

// Spectrum index held by (half h, reg, lane) and the matching slot of the two-wave layout (kernels.hpp).
__host__ __device__ __forceinline__ int spectrum_u_quad(int h, int reg, int lane)
{
    return h + 2 * (lane >> 4) + 8 * ((lane >> 2) & 3) + 32 * (lane & 3) + 128 * reg;
}

// cd bskq[n][2 (p)][2 (h)][L][2 (part)][4 (reg)][64 (lane)]: the 24 KiB a wave reads per step are contiguous.
__host__ __device__ __forceinline__ size_t bskq_index(int L, int i, int p, int h, int l, int part, int reg, int lane)
{
    return (((((size_t)(i * 2 + p) * 2 + h) * L + l) * 2 + part) * 4 + reg) * 64 + lane;
}

// Two-wave layout -> quad layout (same values, permuted); one thread per complex.
static __global__ void k_bsk_quad_from_wave(const cd *__restrict__ src, cd *__restrict__ dst, int n, int L)
{
    const size_t total = (size_t)n * 2 * L * 2 * 512;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, reg = (idx >> 6) & 3, part = (idx >> 8) & 1;
    size_t rest = idx >> 9;
    const int l = rest % L; rest /= L;
    const int h = rest & 1, p = (rest >> 1) & 1;
    const int i = (int)(rest >> 2);
    const int u = spectrum_u_quad(h, reg, lane);
    // kernels.hpp: u = m + 8m' + 64m'' at (reg m'', lane 8m+m')
    dst[idx] = src[bsk_index(L, i, p, l, part, u >> 6, 8 * (u & 7) + ((u >> 3) & 7))];
}

// y_k = sum_n x_n i^(S n k), in place: 16 additions.
template <int S> __device__ __forceinline__ void dft4(cd (&x)[4])
{
    const cd t0 = x[0] + x[2], t1 = x[0] - x[2], t2 = x[1] + x[3], t3 = mul_i<S>(x[1] - x[3]);
    x[0] = t0 + t2; x[2] = t0 - t2; x[1] = t1 + t3; x[3] = t1 - t3;
}

// Pre-twist by (1, w1, w2, w3), then dft4<S>: a forward level.  The products x2*w2 and x3*w3 are never formed: they go
// straight into the first butterfly stage as FMA chains (t0 = x0 + x2 w2: four FMAs) and the difference is recovered as
// 2 x0 - t0 (two FMAs) -- 24 instructions instead of the 12 + 16 of three cmul and a dft4.  The recovered difference
// carries t0's rounding error (half an ulp of the larger of the two), i.e. the transform's error stays of the same order:
// exactness margin at N = 1024, L = 3 is ~0.006 against 0.5, outputs are bit-identical.
template <int S> __device__ __forceinline__ void dft4_pretwist(cd (&x)[4], const cd w1, const cd w2, const cd w3)
{
    const cd t0 = {fma(x[2].re, w2.re, fma(-x[2].im, w2.im, x[0].re)), fma(x[2].re, w2.im, fma(x[2].im, w2.re, x[0].im))};
    const cd t1 = {fma(2.0, x[0].re, -t0.re), fma(2.0, x[0].im, -t0.im)};
    const cd y1 = cmul(x[1], w1);
    const cd t2 = {fma(x[3].re, w3.re, fma(-x[3].im, w3.im, y1.re)), fma(x[3].re, w3.im, fma(x[3].im, w3.re, y1.im))};
    const cd t3 = mul_i<S>(cd{fma(2.0, y1.re, -t2.re), fma(2.0, y1.im, -t2.im)});
    x[0] = t0 + t2; x[2] = t0 - t2; x[1] = t1 + t3; x[3] = t1 - t3;
}

// Exchange 1 WITHOUT LDS.  (reg m; lane 16b + c) -> (reg b; lane 16m + c) is a 4 x 4 transpose of the four registers across
// the wavefront's four 16-lane rows, which gfx950's two swap instructions do directly: v_permlane32_swap exchanges the upper
// 32 lanes of one register with the lower 32 of another (the 2 x 2 block step), v_permlane16_swap the odd rows of one with the
// even rows of another (the step inside the blocks) -- four swaps per 32-bit plane, 16 for four complex doubles, against four
// ds_write_b128 + four ds_read_b128 and their round trip.  The eight-wave kernel's forward phase is bound by the LDS store path
// (profiles/r04_b_oct_floor.txt): this takes a third of its exchange bytes off that path: 2.375 -> 2.20 ms per blind rotate.
// (The other two exchanges permute lanes INSIDE a row, where a register path needs a select per 32-bit plane and stage: exchange 3 as
// 32 v_cndmask_b32_dpp was measured, 2.18 -> 2.40 ms; they stay in LDS.)
__device__ __forceinline__ void swap_halves32(double &a, double &b)
{
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void swap_rows16(double &a, double &b)
{
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a), blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void quad_row_transpose(cd (&x)[4])
{
    swap_halves32(x[0].re, x[2].re); swap_halves32(x[0].im, x[2].im);
    swap_halves32(x[1].re, x[3].re); swap_halves32(x[1].im, x[3].im);
    swap_rows16(x[0].re, x[1].re); swap_rows16(x[0].im, x[1].im);
    swap_rows16(x[2].re, x[3].re); swap_rows16(x[2].im, x[3].im);
}

// LDS slots (16 B each, 256 per wave) of the three exchanges; (hi, mid, lo) = the lane's three base-4 digits.
//   exchange 1: element (reg m; lane b,c,d)      at 64m + 16b + 4c + d
//   exchange 2: element (reg m'; lane m,c,d)     at 64m + 16m' + 4((c+m')&3) + d
//   exchange 3: element (reg m''; lane m,m',d)   at 64m + 16d + 4m' + ((d+m'')&3)
// In both directions every ds_write_b128 pass (8 contiguous lanes, 32 banks of 4 B: 8 slots) touches each slot
// residue mod 8 once and every ds_read_b128 service group (the four 16-lane groups of MI355X_MICROARCH.md, 64
// banks: 16 slots) each residue mod 16 once: SQ_LDS_BANK_CONFLICT = 0 (a first layout of exchange 3 that was only
// checked mod 16 cost 32 conflict cycles per step on the inverse's stores -- profiles/r02_c_pmc128_quad_summary.txt).
struct QuadLane {
    int lane, hi, mid, lo;
};
__device__ __forceinline__ QuadLane quad_lane(int lane) { return {lane, lane >> 4, (lane >> 2) & 3, lane & 3}; }

// NB independent forward transforms advanced level by level through one scratch (in-order DS argument of
// negacyclic_fft.hpp: fft512_forward_batch).  x[t][a] = y[64a + lane] in, spectrum order out.
// (Timing ablations of this kernel -- no exchanges, no barriers, cache-hot key, static priorities -- are recorded in
// profiles/r02_a_quad_ablation.txt; their switches are no longer compiled.)
template <int NB>
__device__ __forceinline__ void fft256_forward_batch(cd (&x)[NB][4], cd *sc, const cd *__restrict__ T, const QuadTwiddles &tw,
                                                     const QuadLane q)
{
#pragma unroll
    for (int t = 0; t < NB; t++) {
        dft4_pretwist<1>(x[t], T[1], T[2], T[3]);
        quad_row_transpose(x[t]);                   // exchange 1 in registers
    }
#pragma unroll
    for (int t = 0; t < NB; t++) {
        dft4_pretwist<1>(x[t], tw.w[0][0], tw.w[0][1], tw.w[0][2]);
        #pragma unroll
        for (int mp = 0; mp < 4; mp++) sc[64 * q.hi + 16 * mp + 4 * ((q.mid + mp) & 3) + q.lo] = x[t][mp];
        wave_lds_order();
        #pragma unroll
        for (int c = 0; c < 4; c++) x[t][c] = sc[64 * q.hi + 16 * q.mid + 4 * ((c + q.mid) & 3) + q.lo];
        wave_lds_order();
    }
#pragma unroll
    for (int t = 0; t < NB; t++) {
        dft4_pretwist<1>(x[t], tw.w[1][0], tw.w[1][1], tw.w[1][2]);
        #pragma unroll
        for (int mpp = 0; mpp < 4; mpp++) sc[64 * q.hi + 16 * q.lo + 4 * q.mid + ((q.lo + mpp) & 3)] = x[t][mpp];
        wave_lds_order();
        #pragma unroll
        for (int d = 0; d < 4; d++) x[t][d] = sc[64 * q.hi + 16 * d + 4 * q.mid + ((d + q.lo) & 3)];
        wave_lds_order();
    }
#pragma unroll
    for (int t = 0; t < NB; t++) {
        dft4_pretwist<1>(x[t], tw.w[2][0], tw.w[2][1], tw.w[2][2]);
    }
}

// Software-pipelined form (see fft512_forward_batch_pipe): transform t's exchange is issued inside transform t+1's
// arithmetic, one DS instruction per kQuadPipeValu VALU instructions.
constexpr int kQuadPipeValu = 6;
template <int NB>
__device__ __forceinline__ void fft256_forward_batch_pipe(cd (&x)[NB][4], cd *sc, const cd *__restrict__ T, const QuadTwiddles &tw,
                                                          const QuadLane q)
{
    auto xchg = [&](int lvl, int t) {
        if (lvl == 1) {
            quad_row_transpose(x[t]);               // exchange 1 in registers
        } else if (lvl == 2) {
#pragma unroll
            for (int mp = 0; mp < 4; mp++) sc[64 * q.hi + 16 * mp + 4 * ((q.mid + mp) & 3) + q.lo] = x[t][mp];
            wave_lds_order();
#pragma unroll
            for (int c = 0; c < 4; c++) x[t][c] = sc[64 * q.hi + 16 * q.mid + 4 * ((c + q.mid) & 3) + q.lo];
        } else {
#pragma unroll
            for (int mpp = 0; mpp < 4; mpp++) sc[64 * q.hi + 16 * q.lo + 4 * q.mid + ((q.lo + mpp) & 3)] = x[t][mpp];
            wave_lds_order();
#pragma unroll
            for (int d = 0; d < 4; d++) x[t][d] = sc[64 * q.hi + 16 * d + 4 * q.mid + ((d + q.lo) & 3)];
        }
        wave_lds_order();
    };
    auto level = [&](int lvl, int t) {
        if (lvl == 1) dft4_pretwist<1>(x[t], T[1], T[2], T[3]);
        else dft4_pretwist<1>(x[t], tw.w[lvl - 2][0], tw.w[lvl - 2][1], tw.w[lvl - 2][2]);
    };
    auto mix = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, kQuadPipeValu, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, kQuadPipeValu, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    __builtin_amdgcn_sched_barrier(0);
    level(1, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int lvl = 1; lvl <= 4; lvl++) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
            if (lvl == 1 && t == 0) continue;
            // the exchange that precedes this piece of arithmetic in the pipeline: previous transform of this level,
            // or the last transform of the previous level
            if (t > 0 && lvl < 4) xchg(lvl, t - 1);
            else if (t == 0) xchg(lvl - 1, NB - 1);
            level(lvl, t);
            if ((t > 0 && lvl < 4) || t == 0) mix();
        }
    }
}

// Inverse (includes the 1/512 of the whole 512-point transform): spectrum order in, x[a] = y_h[64a + lane] / 2 out.
__device__ __forceinline__ void fft256_inverse(cd (&x)[4], cd *sc, const cd *__restrict__ T, const QuadTwiddles &tw, const QuadLane q)
{
    dft4<-1>(x);
#pragma unroll
    for (int d = 1; d < 4; d++) x[d] = cmulc(x[d], tw.w[2][d - 1]);
    #pragma unroll
    for (int d = 0; d < 4; d++) sc[64 * q.hi + 16 * d + 4 * q.mid + ((d + q.lo) & 3)] = x[d];
    wave_lds_order();
    #pragma unroll
    for (int mpp = 0; mpp < 4; mpp++) x[mpp] = sc[64 * q.hi + 16 * q.lo + 4 * q.mid + ((q.lo + mpp) & 3)];
    wave_lds_order();
    dft4<-1>(x);
#pragma unroll
    for (int c = 1; c < 4; c++) x[c] = cmulc(x[c], tw.w[1][c - 1]);
    #pragma unroll
    for (int c = 0; c < 4; c++) sc[64 * q.hi + 16 * q.mid + 4 * ((c + q.mid) & 3) + q.lo] = x[c];
    wave_lds_order();
    #pragma unroll
    for (int mp = 0; mp < 4; mp++) x[mp] = sc[64 * q.hi + 16 * mp + 4 * ((q.mid + mp) & 3) + q.lo];
    wave_lds_order();
    dft4<-1>(x);
#pragma unroll
    for (int b = 1; b < 4; b++) x[b] = cmulc(x[b], tw.w[0][b - 1]);
    quad_row_transpose(x);                          // exchange 1 backwards: the transpose is its own inverse
    dft4<-1>(x);
#pragma unroll
    for (int a = 0; a < 4; a++) x[a] = cmul(x[a], T[4 + a]);
}

// Key slices of one gadget level for one wave: {keep, send} x 4 registers (32 VGPRs).
struct QuadKeys {
    cd keep[4], send[4];
};
__device__ __forceinline__ void load_quad_keys(QuadKeys &K, const cd *__restrict__ key_iphl /* &bskq[i][p][h][l] */, int p, int lane)
{
    const cd *kA = key_iphl + lane, *kB = kA + 256;
    const cd *kKeep = p ? kB : kA, *kSend = p ? kA : kB;       // wave p keeps output p
#pragma unroll
    for (int k = 0; k < 4; k++) {
        K.keep[k] = kKeep[k * 64];
        K.send[k] = kSend[k * 64];
    }
}

// One workgroup = 4*ITEMS waves = ITEMS bootstraps; wave w: item w>>2, polynomial p = (w>>1)&1, half h = w&1.
// WPS = waves per SIMD the launch shape needs (register budget 512 / WPS).  KD = key levels fetched at the top
// of a step: L (all of them, 32 L VGPRs, for one wave per SIMD where nothing else hides the L2 latency) or 1
// (level 0 at the top, level l+1 under the products of level l, as k_blind_rotate does).
template <int L, int BGBIT, int ITEMS, int WPS, int KD = (WPS == 1 ? L : 1)>
__global__ __launch_bounds__(256 * ITEMS, WPS) void k_blind_rotate_quad(BlindRotateArgs A)
{
    constexpr int N = 1024, W = 4 * ITEMS;
    constexpr double r = 0.70710678118654752440;
    __shared__ cd scAll[W][256];            // FFT exchanges
    __shared__ cd sendAll[W][256];          // partner-polynomial hand-over
    __shared__ cd swapAll[W][256];          // half swap
    __shared__ uint32_t accAll[W][N];       // private accumulator copies
    __shared__ uint16_t abarAll[ITEMS][kMaxLweDim];
    __shared__ int btAll[ITEMS];

    const int lane = threadIdx.x & 63, tid = threadIdx.x & 255;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int grp = ITEMS > 1 ? w >> 2 : 0, p = (w >> 1) & 1, h = w & 1;
    uint16_t (&abarL)[kMaxLweDim] = abarAll[grp];
    int &btL = btAll[grp];
    uint32_t *acc = accAll[w];
    cd *sc = scAll[w];
    const int wg_first = A.first + blockIdx.x * ITEMS;
    if (!gate_item_live(A, wg_first)) return;       // list entries past the device-side count (kernels.hpp)
    int item = wg_first + grp;
    const bool live = ITEMS == 1 || (blockIdx.x * ITEMS + grp < A.batch && gate_item_live(A, item));
    if (!live) item = wg_first;                     // the idle group recomputes the first item, stores nothing
    const int n = A.n;

    // ---- gate linear prep + mod-switch (gates_helper.go:10-63, evaluator.go:116,122)
    const bool bad_op = gate_prep_modswitch(A, item, tid, 256, N, abarL, &btL);
    const cd *T = A.twq + (size_t)h * kTwQuadHalf;
    QuadTwiddles tw;
    load_quad_twiddles(tw, T, lane);
    const QuadLane q = quad_lane(lane);
    __syncthreads();

    // ---- acc = X^bt * testvec, this wave's own copy of polynomial p (evaluator.go:117-118, buffer_methods.go:133-164)
    {
        const int bt = btL & (2 * N - 1);
        const uint32_t *tv = A.tv + (size_t)item * A.tv_stride + (size_t)p * N;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = 64 * k + lane;
            const int s = (j - bt) & (2 * N - 1);
            uint32_t v = tv[s & (N - 1)];
            v ^= 0u - (uint32_t)((s >> 10) & 1);      // "negation" is the bitwise complement
            acc[j] = v;
        }
    }
    wave_lds_order();

    constexpr size_t kStep = (size_t)2 * L * 2 * 512;            // cd per CMUX step
    const cd *key = A.bskq + ((size_t)p * 2 + h) * (L * 2 * 256);
    const int partner = w ^ 2, sibling = w ^ 1;
    const double sr = h ? -r : r;                                // (-1)^h / sqrt2
    constexpr bool kSmall = (BGBIT - 1) + 31 + 10 + (L == 1 ? 1 : L == 2 ? 2 : 3) < 51;      // see external_product_core
    const int nsteps = A.nsteps;
    for (int i = 0; i < nsteps; i++) {
        const int at = __builtin_amdgcn_readfirstlane((int)abarL[i]);
        // this step's key slices: issued first, they land under the decomposition and the forward transforms
        QuadKeys K[KD];
#pragma unroll
        for (int l = 0; l < KD; l++) load_quad_keys(K[l], key + (size_t)i * kStep + (size_t)l * 512, p, lane);
        __builtin_amdgcn_sched_barrier(0);
        // d = X^at*acc - acc (evaluator.go:93-96,122-126), decomposed (decomposer.go:55-66) and folded to this
        // half-tree: y_h[j] = dig(z_j) + (-1)^h rho dig(z_{j+256}), z_j = d_j + i d_{j+512}.  rho*(a+ib) =
        // ((a-b) + i(a+b))/sqrt2 with a-b, a+b formed on the integer digits.
        cd x[L][4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            uint32_t dd[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = 64 * a + lane + 256 * k;          // k = 0: re lo, 1: re hi (j+256), 2: im lo (j+512), 3: im hi
                const int s = (j - at) & (2 * N - 1);
                uint32_t v = acc[s & (N - 1)];
                v ^= 0u - (uint32_t)((s >> 10) & 1);
                dd[k] = (v - acc[j] + A.offset) ^ digit_flip_mask<L, BGBIT>();
            }
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int shift = 32 - (l + 1) * BGBIT;
                const int lo_re = digit_of<BGBIT>(dd[0], shift), hi_re = digit_of<BGBIT>(dd[1], shift);
                const int lo_im = digit_of<BGBIT>(dd[2], shift), hi_im = digit_of<BGBIT>(dd[3], shift);
                x[l][a] = cd{fma(sr, (double)(hi_re - hi_im), (double)lo_re), fma(sr, (double)(hi_re + hi_im), (double)lo_im)};
            }
        }
        if constexpr (L > 1) fft256_forward_batch_pipe<L>(x, sc, T, tw, q);
        else fft256_forward_batch<L>(x, sc, T, tw, q);
        cd keep[4], send[4];
#pragma unroll
        for (int l = 0; l < L; l++) {
            const QuadKeys &Kl = K[KD == 1 ? 0 : l];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (l == 0) {
                    keep[k] = cmul(x[0][k], Kl.keep[k]);
                    send[k] = cmul(x[0][k], Kl.send[k]);
                } else {
                    cfma(keep[k], x[l][k], Kl.keep[k]);
                    cfma(send[k], x[l][k], Kl.send[k]);
                }
            }
            if (KD == 1 && l + 1 < L) {
                // anchor the products before the registers are refilled (see external_product_core).  NOT volatile: in a
                // kernel body (no __restrict__ scope around it) a volatile asm counts as a possible store to anything and
                // de-scalarises every wave-uniform twiddle load of the loop (negacyclic_fft.hpp, expand_pow_once)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    asm("" : "+v"(keep[k].re), "+v"(keep[k].im), "+v"(send[k].re), "+v"(send[k].im));
                __builtin_amdgcn_sched_barrier(0);
                load_quad_keys(K[0], key + (size_t)i * kStep + (size_t)(l + 1) * 512, p, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) sendAll[w][k * 64 + lane] = send[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) keep[k] = keep[k] + sendAll[partner][k * 64 + lane];
        fft256_inverse(keep, sc, T, tw, q);
#pragma unroll
        for (int k = 0; k < 4; k++) swapAll[w][k * 64 + lane] = keep[k];
        __syncthreads();
        // undo the radix-2 level: z_j = y0 + y1, z_{j+256} = conj(rho)(y0 - y1) (the 1/2 is in the inverse's scale);
        // own - other = (-1)^h (y0 - y1).  acc += round(.) on the private copy (evaluator.go:102-105).
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const cd o = swapAll[sibling][a * 64 + lane];
            const cd s = keep[a] + o, dl = keep[a] - o;
            const double e1r = (dl.re + dl.im) * sr, e1i = (dl.im - dl.re) * sr;
            const int j = 64 * a + lane;
            const uint32_t e4[4] = {kSmall ? round_to_torus_small(s.re) : round_to_torus_wide(s.re),
                                    kSmall ? round_to_torus_small(e1r) : round_to_torus_wide(e1r),
                                    kSmall ? round_to_torus_small(s.im) : round_to_torus_wide(s.im),
                                    kSmall ? round_to_torus_small(e1i) : round_to_torus_wide(e1i)};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // one ds_add_u32 at two waves per SIMD (-2.2 % at 512 bootstraps); read, add, write for the lone wave (+2.5 %)
                if constexpr (WPS == 2) lds_add(&acc[j + 256 * k], e4[k]);
                else acc[j + 256 * k] += e4[k];
            }
        }
        wave_lds_order();
    }

    if (!live) return;
    // wave (p, h) stores coefficient blocks [256h, 256h+256) and [512+256h, 512+256h+256) of polynomial p
    uint32_t *out = A.out + (size_t)item * 2 * N + (size_t)p * N;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int j = 64 * (k & 3) + lane + 256 * h + 512 * (k >> 2);
        out[j] = acc[j];
    }
    report_bad_op(A, bad_op, tid);
}

// ---------------------------------------------------------------------------------------------------------
// Eight waves per bootstrap (launches of at most one bootstrap per CU): the four (p, h) waves above, twice.
// One group transforms gadget levels [0, L0), the other levels [L0, L) of the same (p, h) half-polynomial, so the
// forward phase of a CMUX step is ceil(L/2) transforms deep instead of L and both waves of a SIMD issue side by
// side.  The group with fewer levels then gathers the four partial products of its output half (its own, the other
// group's, and the partner polynomial's two), runs the one inverse transform and publishes the result.  The accumulator
// is one signed table of 3N words per polynomial, T[s] = acc[s], ~acc[s - N], acc[s - 2N], shared by the polynomial's
// four waves: the decomposition reads X^a*acc - acc through two base addresses and immediate offsets, each wave
// updates its quarter of the coefficients after the half swap, and a third barrier publishes the table.
template <int L, int BGBIT, int LB, int NL, int NK>
__device__ __forceinline__ void oct_forward(const BlindRotateArgs &A, const uint32_t *Tp /* signed table of polynomial p */,
                                            int at, double sr, int lane, const QuadKeys (&K)[NK] /* this step's slices, levels LB.. */,
                                            cd *sc, const cd *__restrict__ T, const QuadTwiddles &tw, const QuadLane q, cd (&keep)[4],
                                            cd (&send)[4], PhaseClock &tr)
{
    constexpr int N = 1024;
    cd x[NL][4];
    // X^at * acc - acc straight from the signed table (T[s] = acc[s], ~acc[s - N], acc[s - 2N] for s in [0, N), [N, 2N),
    // [2N, 3N)): two base addresses per step, every coefficient an immediate offset from them
    // -acc[j] = ~acc[j] + 1 and ~acc[j] is the table's middle copy: rot - own + offset is ONE v_add3_u32
    const uint32_t *rot = Tp + ((lane - at) & (2 * N - 1)), *own_c = Tp + N + lane;
    const uint32_t off1 = A.offset + 1u;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        uint32_t dd[4];
#pragma unroll
        for (int k = 0; k < 4; k++) dd[k] = (rot[64 * a + 256 * k] + own_c[64 * a + 256 * k] + off1) ^ digit_flip_mask<L, BGBIT>();
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const int shift = 32 - (LB + l + 1) * BGBIT;
            const int lo_re = digit_of<BGBIT>(dd[0], shift), hi_re = digit_of<BGBIT>(dd[1], shift);
            const int lo_im = digit_of<BGBIT>(dd[2], shift), hi_im = digit_of<BGBIT>(dd[3], shift);
            x[l][a] = cd{fma(sr, (double)(hi_re - hi_im), (double)lo_re), fma(sr, (double)(hi_re + hi_im), (double)lo_im)};
        }
    }
    tr.mark(0);
    if constexpr (NL > 1) fft256_forward_batch_pipe<NL>(x, sc, T, tw, q);
    else fft256_forward_batch<NL>(x, sc, T, tw, q);
    tr.mark(1);
#pragma unroll
    for (int l = 0; l < NL; l++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (l == 0) {
                keep[k] = cmul(x[0][k], K[0].keep[k]);
                send[k] = cmul(x[0][k], K[0].send[k]);
            } else {
                cfma(keep[k], x[l][k], K[l].keep[k]);
                cfma(send[k], x[l][k], K[l].send[k]);
            }
        }
}


template <int L, int BGBIT>
__global__ __launch_bounds__(512, 1) void k_blind_rotate_oct(BlindRotateArgs A)
{
    static_assert(L >= 2, "one gadget level has nothing to split");
    constexpr int N = 1024, L0 = (L + 1) / 2, L1 = L - L0;
    constexpr double r = 0.70710678118654752440;
    __shared__ cd scAll[8][256];            // FFT exchanges; a group-0 wave also leaves its products for its own half here
    __shared__ cd sendG[2][4][256];         // products for the partner polynomial, by group
    __shared__ cd swapAll[4][256];          // inverse transform results, for the half swap
    __shared__ uint32_t accT[2][3 * N];     // per polynomial: the signed table of the accumulator, T[s] = acc[s], ~acc[s - N],
                                            // acc[s - 2N]: coefficient j of X^a*acc is T[((-a) mod 2N) + j], no wrap, no sign fix
    __shared__ uint16_t abarL[kMaxLweDim];
    __shared__ int btL;

    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // group 1 (the longer chain: it also runs the inverse transform) takes waves 0-3: the older wave of a SIMD wins the
    // issue arbiter (profiles/r02_f_phase_trace.txt)
    const int g = 1 - (w >> 2), ph = w & 3, p = ph >> 1, h = ph & 1;
    const int g0_wave = 4 + ph;
    uint32_t *Tp = accT[p];
    cd *sc = scAll[w];
    const int item = A.first + blockIdx.x;
    if (!gate_item_live(A, item)) return;
    const bool bad_op = gate_prep_modswitch(A, item, tid, 512, N, abarL, &btL);
    const cd *Tg = A.twq + (size_t)h * kTwQuadHalf;
    QuadTwiddles tw;
    load_quad_twiddles(tw, Tg, lane);
    // the eight wave-uniform level-1 twiddles live in scalar registers for the whole kernel (loaded once, here): loaded inside the
    // loop they have to be proven unclobbered on every path, and the s_setprio of the forward phase below is one more thing that
    // defeats that proof -- they would come back as vector loads with a wait each (tests/test_codegen.py counts them)
    cd Tu[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Tu[k] = Tg[k];
    const cd *T = Tu;
    const QuadLane q = quad_lane(lane);
    __syncthreads();
    // the four waves of polynomial p maintain its table together: wave (g, h) owns coefficients 64 qa + lane + 256 k
    const int qa = 2 * g + h;
    {
        const int bt = btL & (2 * N - 1);
        const uint32_t *tv = A.tv + (size_t)item * A.tv_stride + (size_t)p * N;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = 64 * qa + lane + 256 * k;
            const int s = (j - bt) & (2 * N - 1);
            uint32_t v = tv[s & (N - 1)];
            v ^= 0u - (uint32_t)((s >> 10) & 1);       // "negation" is the bitwise complement (buffer_methods.go:152,158)
            Tp[j] = v;
            Tp[j + N] = ~v;
            Tp[j + 2 * N] = v;
        }
    }
    __syncthreads();

    constexpr size_t kStep = (size_t)2 * L * 2 * 512;
    const cd *key = A.bskq + ((size_t)p * 2 + h) * (L * 2 * 256);
    const int partner = ph ^ 2;
    const double sr = h ? -r : r;
    constexpr bool kSmall = (BGBIT - 1) + 31 + 10 + (L == 1 ? 1 : L == 2 ? 2 : 3) < 51;
    const int nsteps = A.nsteps;
    PhaseClock tr;
    tr.start();
    // The key slices of a step are requested during the previous step, where nothing waits for them -- group 1 right after its
    // products (ahead of barrier 1), group 0 right after barrier 1 (it then waits for group 1's inverse) -- so the 8 or 16 loads per
    // wave, their address arithmetic and their queueing at the CU's one address path are off the critical path, and a step starts
    // with its decomposition.  K is one register set for both groups (a wave is in one).
    QuadKeys K[L0];
    const int LBg = g == 0 ? 0 : L0;
    auto request_keys = [&](int step) {
        const cd *kp = key + (size_t)step * kStep + (size_t)LBg * 512;
        if (g == 0) {
#pragma unroll
            for (int l = 0; l < L0; l++) load_quad_keys(K[l], kp + (size_t)l * 512, p, lane);
        } else {
#pragma unroll
            for (int l = 0; l < L1; l++) load_quad_keys(K[l], kp + (size_t)l * 512, p, lane);
        }
    };
    if (nsteps > 0) request_keys(0);
    for (int i = 0; i < nsteps; i++) {
        const int at = __builtin_amdgcn_readfirstlane((int)abarL[i]);
        const int inext = i + 1 < nsteps ? i + 1 : i;          // the last step re-requests its own slices (never used)
        cd keep[4], send[4];
        if (g == 0) {
            // group 0 is the step's critical path (two forward transforms against one): it issues ahead of the group-1 wave it shares
            // its SIMD with, which has ~2,500 cycles of slack before barrier 1.  -3 to -4 % at 128 and 256 bootstraps, +-0 at one
            // (profiles/r04_b_oct_floor.txt; with the twiddles in SGPRs -- as a builtin in round 3's kernel it cost 5 %).
            __builtin_amdgcn_s_setprio(3);
            oct_forward<L, BGBIT, 0, L0>(A, Tp, at, sr, lane, K, sc, T, tw, q, keep, send, tr);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sc[k * 64 + lane] = keep[k];
                sendG[0][ph][k * 64 + lane] = send[k];
            }
            wave_lds_order();       // (*) see below
        } else {
            oct_forward<L, BGBIT, L0, L1>(A, Tp, at, sr, lane, K, sc, T, tw, q, keep, send, tr);
#pragma unroll
            for (int k = 0; k < 4; k++) sendG[1][ph][k * 64 + lane] = send[k];
            request_keys(inext);
            wave_lds_order();       // (*)
        }
        // (*) Every conditional block of this loop ENDS IN A FENCE (no instruction: wavefront scope).  LLVM marks a uniform
        // global load "not clobbered" -- the precondition for s_load -- by walking MemorySSA upwards; at a control-flow
        // merge it tests each incoming definition WITHOUT alias analysis, so a block ending in a plain LDS store makes
        // every wave-uniform twiddle load of the loop a global_load + s_waitcnt vmcnt (eight per step here, four of them
        // on the serial tail).  Fences are skipped by that test, and the walk behind them is alias-aware again.
        tr.mark(2);
        __syncthreads();
        tr.mark(3);
        if (g == 0) {
            request_keys(inext);
            wave_lds_order();       // (*)
        }
        if (g == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                keep[k] = (keep[k] + scAll[g0_wave][k * 64 + lane]) + (sendG[0][partner][k * 64 + lane] + sendG[1][partner][k * 64 + lane]);
            tr.mark(4);
            fft256_inverse(keep, sc, T, tw, q);
#pragma unroll
            for (int k = 0; k < 4; k++) swapAll[ph][k * 64 + lane] = keep[k];
            wave_lds_order();       // (*)
            tr.mark(5);
        }
        __syncthreads();
        tr.mark(6);
        // undo the radix-2 level for this wave's quarter: z_j = y0 + y1, z_{j+256} = conj(rho)(y0 - y1) (the 1/2 is in
        // the inverse's scale), acc += round(.) (evaluator.go:102-105), written to all three copies of the table
        {
            const cd y0 = swapAll[2 * p][qa * 64 + lane], y1 = swapAll[2 * p + 1][qa * 64 + lane];
            const cd s = y0 + y1, dl = y0 - y1;
            const double z[4] = {s.re, (dl.re + dl.im) * r, s.im, (dl.im - dl.re) * r};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = 64 * qa + lane + 256 * k;
                const uint32_t v = Tp[j] + (kSmall ? round_to_torus_small(z[k]) : round_to_torus_wide(z[k]));
                Tp[j] = v;                              // (three ds_add_u32 instead: +1 %)
                Tp[j + N] = ~v;
                Tp[j + 2 * N] = v;
            }
        }
        tr.mark(7);
        __syncthreads();
        tr.mark(8);
    }
#ifdef PHASE_TRACE
    tr.store(A.out + (size_t)item * 2 * N, w, lane);
    return;
#endif

    if (g == 0) {
        // wave (p, h) stores coefficient blocks [256h, 256h+256) and [512+256h, 512+256h+256) of polynomial p
        uint32_t *out = A.out + (size_t)item * 2 * N + (size_t)p * N;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = 64 * (k & 3) + lane + 256 * h + 512 * (k >> 2);
            out[j] = Tp[j];
        }
    }
    report_bad_op(A, bad_op, tid);
}

} // namespace tfhe
