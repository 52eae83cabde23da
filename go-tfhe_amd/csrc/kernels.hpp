// kernels.hpp -- HIP kernels of the gate-bootstrap path (gfx950 / CDNA4).
//
//   k_blind_rotate        evaluator.BlindRotateAssign (evaluator.go:110-135) with the gate's
//                         linear preparation (gates_helper.go:10-63) and the mod-switch fused
//                         into its prologue.  One bootstrap = two wavefronts (wave 0 owns accumulator
//                         polynomial A, wave 1 owns B), 1, 2 or 4 bootstraps per workgroup by launch
//                         size.  The n CMUX steps run inside the kernel with the accumulator resident
//                         in LDS; only the bootstrapping key is streamed.  (Launches of up to one
//                         workgroup per CU use the four-wave form, kernels_quad.hpp; N = 2048 and
//                         N = 512 rings have their own files.)
//   gate_prep_modswitch   the prologue all blind-rotate kernels share, incl. the list indirection of the
//                         device-side MUX split;  k_mux_count / _scan / _fill / _all build that list.
//   k_external_product    evaluator.ExternalProductAssign (evaluator.go:50-81), same core.
//   k_extract_keyswitch   trlwe.SampleExtractIndexAssign + trgsw.IdentityKeySwitchingAssign
//                         (trlwe_ops.go:10-21, keyswitch.go:10-37), per-ciphertext gather (small batches);
//   k_keyswitch_pair / k_keyswitch_wide   the same for batches, key rows shared by a tile of ciphertexts.
//   k_bsk_from_fourier / k_bsk_from_torus / k_ksk_pack   key ingestion into device layouts.
//   k_to_fourier / k_to_poly                              FFT test seams.
#pragma once

#include "negacyclic_fft.hpp"

namespace tfhe {

// ------------------------------------------------------------------------------------
// Device layouts
//
// Bootstrapping key (wave-native): cd bsk[n][2][L][2][8][64]
//     [i]     LWE index / CMUX step
//     [p]     which wave consumes it: p = 0 rows of the A digits, p = 1 rows of the B digits
//             (reference row r = p*L + l, trgsw.go:51-54 / evaluator.go:59-61)
//     [l]     gadget level
//     [part]  0 = A spectrum of the row, 1 = B spectrum
//     [reg][lane]  spectrum order of fft512_forward; one (reg) slice = 64 lanes x 16 B = 1 KiB,
//             so every key load is one fully coalesced global_load_dwordx4 per wave.
//     Same byte count as the reference's [n][2L][2][N] float64 (68,812,800 B at 128-bit).
//
// Key-switching key (packed): uint32 ksk[N][t][base-1][n1p], n1p = (n+1) rounded up to 32 words (whole 128-byte lines).
//     The k = 0 rows of the reference table are all-zero and never read (keyswitch.go:30),
//     so they are not stored; rows are padded so each lane can fetch 16 B aligned.
// ------------------------------------------------------------------------------------

__host__ __device__ __forceinline__ size_t bsk_index(int L, int i, int p, int l, int part, int reg, int lane)
{
    return ((((size_t)(i * 2 + p) * L + l) * 2 + part) * 8 + reg) * 64 + lane;
}

// Gate linear preparation: out = sa*a + sb*b, body += cst (gates_helper.go:10-63,
// gates.go:52-104).  XNOR uses the scalar gate's +1/4 (gates.go:56).
struct GateCoef {
    uint32_t sa, sb, cst;
};

__host__ __device__ __forceinline__ GateCoef gate_coef(int op)
{
    const uint32_t E = 0x20000000u, Q = 0x40000000u, m1 = 0xFFFFFFFFu;
    switch (op) {
    case 0: return {m1, m1, E};          // NAND  -(a+b) + 1/8
    case 1: return {1u, 1u, 0u - E};     // AND    a+b   - 1/8
    case 2: return {1u, 1u, E};          // OR     a+b   + 1/8
    case 3: return {1u, 2u, Q};          // XOR    a+2b  + 1/4
    case 4: return {1u, 0u - 2u, Q};     // XNOR   a-2b  + 1/4
    case 5: return {m1, m1, 0u - E};     // NOR   -(a+b) - 1/8
    case 6: return {m1, 1u, 0u - E};     // ANDNY -a+b   - 1/8
    case 7: return {1u, m1, 0u - E};     // ANDYN  a-b   - 1/8
    case 8: return {m1, 1u, E};          // ORNY  -a+b   + 1/8
    case 9: return {1u, m1, E};          // ORYN   a-b   + 1/8
    default: return {1u, 0u, 0u};        // plain: ct = a
    }
}

struct BlindRotateArgs {
    const cd *bsk;          // device layout above
    const cd *tw;           // twiddle table (negacyclic_fft.hpp)
    const uint32_t *in0;    // [B][n+1]
    const uint32_t *in1;    // [B][n+1] or nullptr (plain bootstrap)
    const uint8_t *ops;     // [B] or nullptr
    int op_uniform;         // used when ops == nullptr; < 0 = plain (ct = in0)
    const uint32_t *tv;     // [2][N] or [B][2][N]
    long tv_stride;         // 0 or 2N
    uint32_t *out;          // [B][2][N]
    int n, nsteps, Nbit;
    uint32_t offset;        // decomposition offset (cloudkey.go:60-71)
    int first, batch;       // this launch covers items [first, first + batch) of the arrays above
    const cd *bskq;         // the same key in the four-wave layout (kernels_quad.hpp), N = 1024 shapes only
    const cd *twq;          // twiddle table of the four-wave kernel
    // Device-side MUX split (gates.go:107-114; tfhe_hip.hip: gate_batch_device).  With idx != nullptr, item
    // v < split is a direct item whose MUX op code reads as AND (pass A: AND(a,b)); item v >= split is entry
    // k = v - split of the compact list idx[0 .. *count): first operand row idx[k] of in0, second operand row
    // (list_in1_by_idx ? idx[k] : k) of list_in1, gate list_op.  Entries k >= *count do not exist: their
    // workgroups exit at once (the host sizes launches for the worst case and never reads the count).
    const int *idx;
    const int *count;
    int split;
    const uint32_t *list_in1;
    int list_in1_by_idx, list_op;
    int *status;            // device word: bit 0 is set when a gate item carries an op code the launch cannot run
};
constexpr int kStatusBadOp = 1;

// Whether item v of a launch exists (always, unless it is a list entry past the device-side count).
__device__ __forceinline__ bool gate_item_live(const BlindRotateArgs &A, int v)
{
    return !A.idx || v < A.split || v - A.split < *A.count;
}

// Prologue shared by every blind-rotate kernel: gate linear preparation (gates_helper.go:10-63) and mod-switch of
// the n+1 words of item v (evaluator.go:116,122) by threads tid, tid + nthreads, ... into abar[0..n) and bt.
// Returns whether the item's op code is one the launch cannot run; the caller reports it with
// report_bad_op() AFTER its CMUX loop: a global atomic ahead of the loop makes every later load "possibly
// clobbered", and the wave-uniform twiddle loads then stop being scalar loads (measured: 6.3 -> 7.7 ms).
__device__ __forceinline__ bool gate_prep_modswitch(const BlindRotateArgs &A, int v, int tid, int nthreads, int N,
                                                    uint16_t *abar, int *bt)
{
    const int n = A.n;
    size_t r0 = (size_t)v, r1 = (size_t)v;
    const uint32_t *p1 = A.in1;
    int op = A.ops ? (int)A.ops[v < A.split || !A.idx ? v : 0] : A.op_uniform;
    if (A.idx) {
        if (v < A.split) {
            if (op == 10) op = 1;                       // TFHE_OP_MUX -> TFHE_OP_AND on (a, b)
        } else {
            const int k = v - A.split;
            r0 = (size_t)A.idx[k];
            r1 = A.list_in1_by_idx ? r0 : (size_t)k;
            p1 = A.list_in1;
            op = A.list_op;
        }
    }
    // a two-operand launch only knows the ten binary gates (MUX exists as the passes of the list form):
    // anything else is recorded for tfhe_ctx_sync and runs as a plain bootstrap of the first operand
    const bool bad = p1 && (op < 0 || op > 9);
    const GateCoef g = gate_coef(p1 ? op : -1);
    const uint32_t *x0 = A.in0 + r0 * (n + 1);
    const uint32_t *x1 = p1 ? p1 + r1 * (n + 1) : x0;
    const int sh = 32 - A.Nbit - 1;
    const uint32_t rnd = 1u << (sh - 1);
    for (int x = tid; x <= n; x += nthreads) {
        uint32_t w = g.sa * x0[x] + (p1 ? g.sb * x1[x] : 0u);
        if (x == n) {
            w += g.cst;
            *bt = 2 * N - (int)(((unsigned long long)w + rnd) >> sh);     // int add, no 32-bit wrap (evaluator.go:116)
        } else {
            abar[x] = (uint16_t)((uint32_t)(w + rnd) >> sh);               // wraps (evaluator.go:122)
        }
    }
    return bad;
}

__device__ __forceinline__ void report_bad_op(const BlindRotateArgs &A, bool bad, int tid)
{
    if (bad && tid == 0 && A.status) atomicOr(A.status, kStatusBadOp);
}

constexpr int kMaxLweDim = 1280;      // Uint7/8 use n = 1160 (params.go:444-510)

// Key slices of one gadget level for one wave: 8 register-slices of the spectrum it keeps and
// 8 of the spectrum it hands to its partner (64 VGPRs), fetched one level ahead of use so the
// L2/MALL latency hides under the forward FFT in between.
// acc[j] += v on an LDS word only this wave touches: one ds_add_u32 instead of read, add, write (-0.4 % at 1,024 bootstraps)
__device__ __forceinline__ void lds_add(uint32_t *p, uint32_t v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Phase clock of the blind-rotate kernels (tools/phase_trace.py; -DPHASE_TRACE builds only): per-wave sums of the
// shader clock between marks, stored over the kernel's output at the end.  A mark drains the wave's LDS operations.
struct PhaseClock {
#ifdef PHASE_TRACE
    long long sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prev = 0;
    __device__ __forceinline__ void start() { __builtin_amdgcn_sched_barrier(0); prev = clock64(); __builtin_amdgcn_sched_barrier(0); }
    // with -DPHASE_TRACE_LIGHT only the marks around the first barrier of a step of the two-wave kernel remain (2, 3:
    // the wave has drained its LDS operations there anyway) and everything else is lumped into mark 7: the
    // timeline of an almost undisturbed kernel
    __device__ __forceinline__ void mark(int k)
    {
#ifdef PHASE_TRACE_LIGHT
        if (k != 2 && k != 3 && k != 7) return;
#endif
        __builtin_amdgcn_sched_barrier(0);
        const long long now = clock64();
        sum[k] += now - prev;
        prev = now;
        __builtin_amdgcn_sched_barrier(0);
    }
    // wave `slot` of the item stores its sums into the item's output row (the result is lost: trace builds only)
    __device__ __forceinline__ void store(uint32_t *out_item, int slot, int lane) const
    {
        if (lane < 10) {
            long long v = 0;
#pragma unroll
            for (int k = 0; k < 10; k++) v = lane == k ? sum[k] : v;
            reinterpret_cast<long long *>(out_item)[slot * 16 + lane] = v;
        }
    }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
#endif
};

struct KeyRegs {
    cd keep[8], send[8];
};

__device__ __forceinline__ void load_keys(KeyRegs &K, const cd *__restrict__ key_ipl /* &bsk[i][p][l] */, int p, int lane)
{
    const cd *kA = key_ipl + lane;            // part 0: A spectrum of the row
    const cd *kB = key_ipl + 512 + lane;      // part 1: B spectrum
    // wave p keeps output p: p = 0 accumulates the A output from part 0, p = 1 the B output
    const cd *kKeep = p ? kB : kA;
    const cd *kSend = p ? kA : kB;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        K.keep[k] = kKeep[k * 64];
        K.send[k] = kSend[k * 64];
    }
}

// Source of the polynomial to decompose: either X^at*acc - acc read from this wave's LDS
// accumulator (blind rotate; re-read per gadget level so the 16 words need not stay in VGPRs
// across the transforms), or a plain polynomial in global memory (external-product seam).
struct DiffSource {
    const uint32_t *accL;   // LDS accumulator of this wave's polynomial, or nullptr
    int at;                 // rotation amount in [0, 2N)
    const uint32_t *plain;  // global polynomial when accL == nullptr
};

__device__ __forceinline__ uint32_t diff_coeff(const DiffSource &S, int j)
{
    constexpr int N = 1024;
    if (!S.accL) return S.plain[j];
    const int s = (j - S.at) & (2 * N - 1);
    uint32_t v = S.accL[s & (N - 1)];
    v ^= 0u - (uint32_t)((s >> 10) & 1);          // "negation" is the bitwise complement
    return v - S.accL[j];                          // d = X^at*acc - acc (evaluator.go:93-96,122-126)
}

// One external product, this wave's half: from the polynomial S describes (16 coefficients per
// lane: j = 64a+lane and j+512, a < 8) produce this wave's half of bsk[i] (x) d as 16 torus words.
// Two waves cooperate: each transforms the L digit polynomials of its own polynomial, multiplies
// them with its L key rows into partial sums for BOTH outputs, hands the partner's partial sum
// over through LDS, and inverse-transforms its own.
// (evaluator.go:50-81; decomposer.go:55-66; fourier_ops.go:167-191)
// PRIO: issue priorities by phase (s_setprio) for the launch shape with two FREE-RUNNING workgroups per CU, where the two
// waves of a SIMD belong to different workgroups and drift through different phases: decomposition 3, forward
// transforms 2, products + hand-over 0, gather + inverse transform 1, rounding + accumulator update 3.  The arbiter then
// favours the wave that is in its LDS-exchange-heavy transforms over the one issuing plain fp64 products, and the
// workgroups stay out of step instead of queueing at the LDS pipe together.  What matters is forward > inverse >
// products (equal priorities for the two transforms: 6.27 ms; products = inverse: 5.81; profiles/r03_n_phase_priorities.txt).
template <int L, int BGBIT, bool ALT = false, bool PRIO = false>
__device__ __forceinline__ void external_product_core(const DiffSource &S, uint32_t (&e)[16],
                                                      const cd *__restrict__ key_ip, /* &bsk[i][p] */
                                                      KeyRegs &K, /* scratch: the level's key slices */
                                                      cd *sc_mine, const cd *sc_other,
                                                      const cd *__restrict__ table, const LaneTwiddles &tw,
                                                      uint32_t offset, int p, int lane, PhaseClock &clk)
{
    cd keep[8], send[8];
    // All L digit polynomials are transformed as one batch (fft512_forward_batch), then
    // multiplied into the two accumulators level by level.
    if constexpr (PRIO) TFHE_PRIO(3);
    cd x[L][8];
    {
#pragma unroll
        for (int a = 0; a < 8; a++) {
            constexpr uint32_t flip = digit_flip_mask<L, BGBIT>();
            const uint32_t d0 = (diff_coeff(S, 64 * a + lane) + offset) ^ flip, d1 = (diff_coeff(S, 64 * a + lane + 512) + offset) ^ flip;
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int shift = 32 - (l + 1) * BGBIT;
                x[l][a] = cd{(double)digit_of<BGBIT>(d0, shift), (double)digit_of<BGBIT>(d1, shift)};
            }
        }
    }
    clk.mark(0);
    if constexpr (PRIO) TFHE_PRIO(2);
    // level-0 key slices are requested here, under the last level of the forward transforms (~260 fp64
    // instructions), instead of a whole step ahead: 64 VGPRs free during the inverse transform and levels 1-2
    if constexpr (L > 1) fft512_forward_batch_pipe<L>(x, sc_mine, table, tw, lane, [&] {
        __builtin_amdgcn_sched_barrier(0);
        load_keys(K, key_ip, p, lane);
        __builtin_amdgcn_sched_barrier(0);
    });
    else { load_keys(K, key_ip, p, lane); fft512_forward_batch<L>(x, sc_mine, table, tw, lane); }
    clk.mark(1);
    if constexpr (PRIO) TFHE_PRIO(0);
#pragma unroll
    for (int l = 0; l < L; l++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (l == 0) {
                keep[k] = cmul(x[0][k], K.keep[k]);
                send[k] = cmul(x[0][k], K.send[k]);
            } else {
                cfma(keep[k], x[l][k], K.keep[k]);
                cfma(send[k], x[l][k], K.send[k]);
            }
        }
        // Anchor the accumulators: pure arithmetic has no ordering against the loads below, and
        // without this the compiler sinks the MAC past them (keeping the spectrum alive and spilling
        // the prefetched keys).  Then refill the key registers for the next level / next CMUX step.
#pragma unroll
        for (int k = 0; k < 8; k++) {
            asm volatile("" : "+v"(keep[k].re), "+v"(keep[k].im), "+v"(send[k].re), "+v"(send[k].im));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (l + 1 < L) load_keys(K, key_ip + (size_t)(l + 1) * 2 * 512, p, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    // hand the partner's partial sum over
#pragma unroll
    for (int k = 0; k < 8; k++) sc_mine[k * 64 + lane] = send[k];
    clk.mark(2);
    __syncthreads();
    clk.mark(3);
    if constexpr (PRIO) TFHE_PRIO(1);
#pragma unroll
    for (int k = 0; k < 8; k++) keep[k] = keep[k] + sc_other[k * 64 + lane];
    clk.mark(4);
    if constexpr (ALT) {
        // ONE barrier per step: the inverse transform runs in the PARTNER's scratch, whose content (the partner's products
        // for this wave) this very wave has just consumed, and the caller swaps the two buffers' roles every step -- a
        // buffer is then only ever touched by the wave that used it last, until the next barrier hands it over.
        // 5.93 -> 5.84 ms at 1,024 bootstraps (then four per workgroup), nothing at 512, +3 % at 768 (one per workgroup):
        // used by the full-launch shape only (profiles/r03_d_headline_variants.txt).
        fft512_inverse_pipe(keep, const_cast<cd *>(sc_other), table, tw, lane);
    } else {
        __syncthreads();
        clk.mark(5);
        fft512_inverse_pipe(keep, sc_mine, table, tw, lane);
    }
    // |v| <= 2L * N * (Bg/2) * 2^31: below 2^51 the 1.5*2^52 trick is exact (L=3, Bgbit=6: 2^48.6);
    // the Uint1 / Uint3 shapes (L=2,Bgbit=10: 2^52; L=1,Bgbit=23: 2^64) need the wide form and sit in
    // the tolerance regime, like the reference's own fp64 pipeline at those sets.
    if constexpr (PRIO) TFHE_PRIO(3);
    constexpr bool kSmall = (BGBIT - 1) + 31 + 10 + (L == 1 ? 1 : L == 2 ? 2 : 3) < 51;
#pragma unroll
    for (int a = 0; a < 8; a++) {
        e[a] = kSmall ? round_to_torus_small(keep[a].re) : round_to_torus_wide(keep[a].re);
        e[a + 8] = kSmall ? round_to_torus_small(keep[a].im) : round_to_torus_wide(keep[a].im);
    }
    clk.mark(6);
}

// ITEMS = 2 puts two bootstraps (four waves) in one workgroup.  The hardware places the waves of ONE
// workgroup on distinct SIMDs but not those of two co-resident 2-wave workgroups (tools/ubench_placement.hip:
// two workgroups per CU land as [2 0 1 1] waves per SIMD, one SIMD idle), so launches of 1..2 workgroups per
// CU use this form: one 4-wave workgroup per CU = [1 1 1 1].  The two items only share the barriers.
// (Three waves per SIMD do not fit this kernel's registers: profiles/r02_e_occupancy.txt.)
//
// FULL = the launch puts two waves from DIFFERENT workgroups on a SIMD: 769...1,024 bootstraps as two four-wave workgroups
// per CU (ITEMS = 2: the one-barrier step ALT and the phase priorities PRIO of external_product_core), 513...768 as three
// two-wave workgroups per CU (ITEMS = 1: the priorities only -- ALT measured the same there): 5.26 -> 4.57 ms at 768.
// Until late in round 3 the full launch ran FOUR bootstraps in one eight-wave workgroup, all in step on the key stream
// (5.84 ms); two free-running four-wave workgroups were slower without priorities (6.27) and are faster with them:
// 5.87 -> 5.33 ms interleaved on one box (-9 %; profiles/r03_n_phase_priorities.txt).
template <int L, int BGBIT, int ITEMS = 1, bool FULL = false>
__global__ __launch_bounds__(128 * ITEMS, 2) void k_blind_rotate(BlindRotateArgs A)
{
    constexpr int N = 1024;
    __shared__ cd scAll[ITEMS][2][kScratchSlots];
    __shared__ uint32_t accAll[ITEMS][2][N];
    __shared__ uint16_t abarAll[ITEMS][kMaxLweDim];
    __shared__ int btAll[ITEMS];

    const int lane = threadIdx.x & 63, tid = threadIdx.x & 127;              // tid: thread within the item's wave pair
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int p = w & 1, grp = ITEMS > 1 ? w >> 1 : 0;       // grp < ITEMS
    cd (&sc)[2][kScratchSlots] = scAll[grp];
    uint32_t (&accL)[2][N] = accAll[grp];
    uint16_t (&abarL)[kMaxLweDim] = abarAll[grp];
    int &btL = btAll[grp];
    // items of this workgroup: [first + blockIdx.x*ITEMS, ...); the live ones form a prefix (ragged batch, or
    // list entries past the device-side count): idle pairs recompute the workgroup's first item and store nothing
    const int wg_first = A.first + blockIdx.x * ITEMS;
    if (!gate_item_live(A, wg_first)) return;
    int item = wg_first + grp;
    const bool live = ITEMS == 1 || (blockIdx.x * ITEMS + grp < A.batch && gate_item_live(A, item));
    if (!live) item = wg_first;
    const int n = A.n;

    // ---- gate linear prep + mod-switch (gates_helper.go:10-63, evaluator.go:116,122)
    const bool bad_op = gate_prep_modswitch(A, item, tid, 128, N, abarL, &btL);
    LaneTwiddles tw;
    load_lane_twiddles(tw, A.tw, lane);
    __syncthreads();

    // ---- acc = X^bt * testvec  (evaluator.go:117-118, buffer_methods.go:133-164)
    // The accumulator lives in LDS (accL[p], this wave's polynomial) between steps.
    {
        const int bt = btL & (2 * N - 1);
        const uint32_t *tv = A.tv + (size_t)item * A.tv_stride + (size_t)p * N;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = 64 * q + lane;
            const int s = (j - bt) & (2 * N - 1);
            uint32_t v = tv[s & (N - 1)];
            v ^= 0u - (uint32_t)((s >> 10) & 1);      // "negation" is the bitwise complement
            accL[p][j] = v;
        }
    }
    wave_lds_order();

    const cd *key = A.bsk + (size_t)p * L * 2 * 512;
    constexpr size_t kStep = (size_t)2 * L * 2 * 512;        // cd elements per CMUX step
    const int nsteps = A.nsteps;
    KeyRegs K;
    PhaseClock clk;
    clk.start();
    for (int i = 0; i < nsteps; i++) {
        const int at = __builtin_amdgcn_readfirstlane((int)abarL[i]);
        uint32_t e[16];
        const DiffSource S{accL[p], at, nullptr};
        constexpr bool kAlt = FULL && ITEMS == 2;         // the one-barrier step: see external_product_core (ALT)
        const int mine = kAlt ? p ^ (i & 1) : p;
        external_product_core<L, BGBIT, kAlt, FULL>(S, e, key + (size_t)i * kStep, K, sc[mine], sc[mine ^ 1], A.tw, tw, A.offset, p, lane, clk);
        // acc += e   (evaluator.go:102-105)
#pragma unroll
        for (int q = 0; q < 16; q++) lds_add(&accL[p][64 * q + lane], e[q]);
        wave_lds_order();
        clk.mark(7);
    }

    if (!live) return;
#ifdef PHASE_TRACE
    clk.store(A.out + (size_t)item * 2 * N, p, lane);
    return;
#endif
    uint32_t *out = A.out + (size_t)item * 2 * N + (size_t)p * N;
#pragma unroll
    for (int q = 0; q < 16; q++) out[64 * q + lane] = accL[p][64 * q + lane];
    report_bad_op(A, bad_op, tid);
}

// ExternalProductAssign of in[b] with bsk[key_index] (test seam).
template <int L, int BGBIT>
__global__ __launch_bounds__(128, 2) void k_external_product(const cd *bsk, const cd *twt, int key_index,
                                                           const uint32_t *in, uint32_t *out, uint32_t offset)
{
    constexpr int N = 1024;
    __shared__ cd sc[2][kScratchSlots];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
    LaneTwiddles tw;
    load_lane_twiddles(tw, twt, lane);
    const uint32_t *src = in + (size_t)blockIdx.x * 2 * N + (size_t)p * N;
    uint32_t e[16];
    const DiffSource S{nullptr, 0, src};
    const cd *key = bsk + ((size_t)key_index * 2 + p) * L * 2 * 512;
    KeyRegs K;
    PhaseClock clk;
    external_product_core<L, BGBIT>(S, e, key, K, sc[p], sc[p ^ 1], twt, tw, offset, p, lane, clk);
    uint32_t *dst = out + (size_t)blockIdx.x * 2 * N + (size_t)p * N;
#pragma unroll
    for (int q = 0; q < 16; q++) dst[64 * q + lane] = e[q];
}

// ------------------------------------------------------------------------------------
// Key ingestion
// ------------------------------------------------------------------------------------

// Reference Fourier layout [n][2L][2][N] float64 -> device layout.  One thread per complex.
static __global__ void k_bsk_from_fourier(const double *__restrict__ src, cd *__restrict__ dst, int n, int L)
{
    const size_t total = (size_t)n * 2 * L * 2 * 512;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, reg = (idx >> 6) & 7, part = (idx >> 9) & 1;
    size_t rest = idx >> 10;
    const int l = rest % L; rest /= L;
    const int p = rest & 1;
    const int i = (int)(rest >> 1);
    const int r = p * L + l;
    const double *poly = src + (((size_t)i * 2 * L + r) * 2 + part) * 1024;
    const int s = reference_slot_1024(reg, lane);
    const int base = 8 * (s >> 2) + (s & 3);
    dst[idx] = cd{poly[base], poly[base + 4]};
}

// Coefficient-domain key [n][2L][2][N] uint32 -> device layout (own forward FFT; replaces
// trgsw.NewTRGSWLv1FFT, trgsw.go:71-82).  One wave per polynomial.
static __global__ __launch_bounds__(64) void k_bsk_from_torus(const uint32_t *__restrict__ src, cd *__restrict__ dst,
                                                        const cd *__restrict__ twt, int L)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    const int polyIdx = blockIdx.x;                       // ((i*2L + r)*2 + part)
    const int part = polyIdx & 1, r = (polyIdx >> 1) % (2 * L), i = (polyIdx >> 1) / (2 * L);
    const int p = r / L, l = r % L;
    LaneTwiddles tw;
    load_lane_twiddles(tw, twt, lane);
    const uint32_t *poly = src + (size_t)polyIdx * 1024;
    cd x[8];
#pragma unroll
    for (int a = 0; a < 8; a++)
        x[a] = cd{(double)(int32_t)poly[64 * a + lane], (double)(int32_t)poly[64 * a + lane + 512]};
    fft512_forward(x, sc, twt, tw, lane);
#pragma unroll
    for (int k = 0; k < 8; k++) dst[bsk_index(L, i, p, l, part, k, lane)] = x[k];
}

// Reference KSK [N*t*base][n+1] -> packed [N*t*(base-1) + 1][n1p] (drops the all-zero k = 0 rows;
// ONE all-zero row is kept at the end as padding target for the unrolled gather).
static __global__ void k_ksk_pack(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int n1, int n1p,
                           int base, size_t rows_packed)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows_packed * (size_t)n1p) return;
    const size_t row = idx / n1p;
    const int x = (int)(idx % n1p);
    const size_t ij = row / (base - 1);
    const int k = (int)(row % (base - 1)) + 1;
    dst[idx] = (x < n1 && row + 1 < rows_packed) ? src[(ij * base + k) * (size_t)n1 + x] : 0u;
}

// ------------------------------------------------------------------------------------
// Sample extract (index 0) + identity key switch.  One workgroup (4 waves) per ciphertext.
//   phase 1: all N*t digits are computed and the non-zero ones compacted into an LDS list
//            (the subtraction is commutative mod 2^32, so order is free);
//   phase 2: each wave walks a quarter of the list; a wave covers one packed row with
//            CH coalesced 16-byte loads per lane and keeps CH uint4 partial sums;
//   phase 3: the four partial sums are combined through LDS.
// ------------------------------------------------------------------------------------
struct KeySwitchArgs {
    const uint32_t *trlwe;   // [B][2][N]
    const uint32_t *ksk;     // packed
    uint32_t *out;           // [B][n+1]
    int n, N, t, basebit, n1p;
    const int *count;        // optional device-side item count (list launches): only min(B, *count) items exist
};
__device__ __forceinline__ int ks_items(const KeySwitchArgs &A, int B)
{
    if (!A.count) return B;
    const int c = *A.count;
    return c < B ? c : B;
}

// `parts` > 1 (few ciphertexts, many idle CUs): workgroup (item, part) sums the rows of digit range `part` of the item and adds its
// partial sum to the output k_ks_init prepared (32-bit atomics: subtraction mod 2^32 is associative and commutative, so the result is
// the same word for word) -- one ciphertext of the Uint5 set is 6,144 rows = 26 MB through ONE CU otherwise (0.34 ms; 16 parts: 0.04).
template <int CH, typename IdxT>
__global__ __launch_bounds__(256) void k_extract_keyswitch(KeySwitchArgs A, int parts)
{
    constexpr int U = 8;                      // key rows in flight per wave
    __shared__ IdxT rows[9216 + 4 * U];
    __shared__ uint32_t red[4][CH * 256];
    __shared__ int count;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int N = A.N, t = A.t, bb = A.basebit, base1 = (1 << bb) - 1;
    const int item = parts > 1 ? (int)blockIdx.x / parts : (int)blockIdx.x, part = parts > 1 ? (int)blockIdx.x - item * parts : 0;
    if (A.count && item >= *A.count) return;
    const uint32_t *ta = A.trlwe + (size_t)item * 2 * N;
    if (tid == 0) count = 0;
    __syncthreads();
    const uint32_t prec = 1u << (32 - (1 + bb * t));
    const int idx_lo = (int)((long long)part * N * t / parts), idx_hi = (int)((long long)(part + 1) * N * t / parts);
    for (int idx = idx_lo + tid; idx < idx_hi; idx += 256) {
        const int i = idx / t, j = idx - i * t;
        // SampleExtractIndexAssign(.,0,.): P[0] = A[0], P[i] = ~A[N-i]  (trlwe_ops.go:13-19)
        const uint32_t ai = i == 0 ? ta[0] : ~ta[N - i];
        const uint32_t k = ((ai + prec) >> (32 - (j + 1) * bb)) & (uint32_t)base1;
        if (k) rows[atomicAdd(&count, 1)] = (IdxT)((uint32_t)idx * base1 + (k - 1));
    }
    __syncthreads();
    const int cnt = count;
    // pad the list to a multiple of 4*U with the all-zero row appended to the packed key
    const uint32_t zero_row = (uint32_t)N * t * base1;
    if (tid < 4 * U) rows[cnt + tid] = (IdxT)zero_row;
    __syncthreads();
    uint4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = make_uint4(0, 0, 0, 0);
    const int quads = A.n1p >> 2;
    for (int e = w; e < cnt; e += 4 * U) {
        uint4 v[U][CH];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint4 *row = reinterpret_cast<const uint4 *>(A.ksk + (size_t)rows[e + 4 * u] * A.n1p);
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int qd = c * 64 + lane;
                v[u][c] = qd < quads ? row[qd] : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) {
                acc[c].x += v[u][c].x; acc[c].y += v[u][c].y; acc[c].z += v[u][c].z; acc[c].w += v[u][c].w;
            }
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
        uint32_t *r = &red[w][(c * 64 + lane) * 4];
        r[0] = acc[c].x; r[1] = acc[c].y; r[2] = acc[c].z; r[3] = acc[c].w;
    }
    __syncthreads();
    uint32_t *out = A.out + (size_t)item * (A.n + 1);
    for (int x = tid; x <= A.n; x += 256) {
        const uint32_t sum = red[0][x] + red[1][x] + red[2][x] + red[3][x];
        if (parts > 1) { if (sum) atomicAdd(out + x, 0u - sum); }
        else out[x] = (x == A.n ? ta[N] : 0u) - sum;      // out = (0,...,0,b) - sum rows (keyswitch.go:18-33)
    }
}

// ------------------------------------------------------------------------------------
// Tiled key switches (k_keyswitch_pair for base 4, k_keyswitch_wide for the larger bases): a tile of
// ciphertexts shares the candidate key rows of each (i, j), so a row crosses L2/MALL once per tile instead of
// once per ciphertext that needs it (the per-ciphertext gather above).  Partial sums of the N/IC coefficient
// ranges are combined with 32-bit atomic adds into `out`, which k_ks_init has set to (0, ..., 0, b)
// (keyswitch.go:18-21).
// ------------------------------------------------------------------------------------
static __global__ void k_ks_init(const uint32_t *__restrict__ trlwe, uint32_t *__restrict__ out, int n, int N, int B,
                                 const int *__restrict__ count)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (count && *count < B) B = *count;
    if (idx >= (size_t)B * (n + 1)) return;
    const int b = (int)(idx / (n + 1)), x = (int)(idx % (n + 1));
    out[idx] = x == n ? trlwe[(size_t)b * 2 * N + N] : 0u;
}

// Base-4 key switch, two (i, j) steps per LDS read.  One single-wave workgroup = T ciphertexts x 256 output
// columns (4 per lane) x IC extracted coefficients.  LDS is used as a per-lane, dynamically indexed register
// file: for each PAIR of consecutive steps the lane builds the 16 sums r[k1] + r'[k2] of its own column quad
// (r[0] = r'[0] = 0; rows straight from L2, one pair ahead) and every ciphertext then needs ONE ds_read_b128
// and one subtraction for two digits -- its two adjacent 2-bit digits are one 4-bit field of a 64-bit scalar
// holding the 2t digits of a coefficient pair.  Lanes only ever read back what they wrote themselves, so there
// is no barrier.  grid = ct_tiles * col_slices * N/IC workgroups, XCD-aware decode (the ciphertext tiles that
// share key rows run on one XCD).  IC even.  (keyswitch.go:10-37, trlwe_ops.go:10-21; subtraction mod 2^32 is
// associative and commutative, so pre-summing rows is exact.)
template <int T>
__global__ __launch_bounds__(64) void k_keyswitch_pair(KeySwitchArgs A, int B, int IC, int ct_tiles, int col_slices)
{
    __shared__ uint4 comb[16][64];
    const int lane = threadIdx.x;
    B = ks_items(A, B);
    const int N = A.N, t = A.t;                       // basebit = 2
    int tx, ty;
    {
        const int id = blockIdx.x, ny = col_slices * (N / IC);
        if ((ny & 7) == 0) { const int slot = id >> 3; tx = slot % ct_tiles; ty = (slot / ct_tiles) * 8 + (id & 7); }
        else { tx = id % ct_tiles; ty = id / ct_tiles; }
    }
    const int b0 = tx * T, q0 = (ty % col_slices) * 64 + lane, i0 = (ty / col_slices) * IC;
    if (b0 >= B) return;
    const int quads = A.n1p >> 2;
    const bool active = q0 < quads;
    const uint32_t prec = 1u << (32 - (1 + 2 * t));
    const int wshift = 32 - 2 * t;
    auto digit_word = [&](int i) -> uint32_t {          // lane b < T holds ciphertext b0 + b
        const int b = b0 + lane;
        if (lane >= T || b >= B || i >= i0 + IC) return 0u;
        const uint32_t *ta = A.trlwe + (size_t)b * 2 * N;
        const uint32_t ai = i == 0 ? ta[0] : ~ta[N - i];                // trlwe_ops.go:13-19
        return (ai + prec) >> wshift;                                     // t digits, most significant first
    };
    const uint4 *kbase = reinterpret_cast<const uint4 *>(A.ksk) + (size_t)i0 * t * 3 * quads + (active ? q0 : 0);
    const size_t rowq = (size_t)quads;
    const int F = IC * t;                                // steps of this workgroup (even)
    uint4 n[6];
    auto load_pair = [&](int f) {                        // rows of steps f and f+1
        const uint4 *rp = kbase + (size_t)(f < F ? f : F - 2) * 3 * rowq;
#pragma unroll
        for (int r = 0; r < 6; r++) n[r] = rp[r * rowq];
    };
    load_pair(0);
    comb[0][lane] = make_uint4(0, 0, 0, 0);
    uint4 acc[T];
#pragma unroll
    for (int b = 0; b < T; b++) acc[b] = make_uint4(0, 0, 0, 0);
    auto add4 = [](uint4 x, uint4 y) { return make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); };
    uint32_t wA = digit_word(i0), wB = digit_word(i0 + 1);
    for (int cp = 0, f = 0; cp < IC / 2; cp++) {
        // lane b: the 2t digits of ciphertext b for this coefficient pair, most significant first
        const unsigned long long W = ((unsigned long long)wA << (2 * t)) | wB;
        wA = digit_word(i0 + 2 * cp + 2);
        wB = digit_word(i0 + 2 * cp + 3);
        for (int P = 0; P < t; P++, f += 2) {
            comb[1][lane] = n[3]; comb[2][lane] = n[4]; comb[3][lane] = n[5];
#pragma unroll
            for (int k1 = 1; k1 < 4; k1++) {
                comb[4 * k1][lane] = n[k1 - 1];
#pragma unroll
                for (int k2 = 1; k2 < 4; k2++) comb[4 * k1 + k2][lane] = add4(n[k1 - 1], n[2 + k2]);
            }
            load_pair(f + 2);                            // lands under the T reads below
            // per-lane byte offset of this ciphertext's table row for the two digits of steps f, f+1
            const int sel = (int)((uint32_t)(W >> (4 * (t - 1 - P))) & 15u) * (int)sizeof(comb[0]);
            const char *mine = reinterpret_cast<const char *>(&comb[0][lane]);
#pragma unroll
            for (int b = 0; b < T; b++) {
                const uint4 r = *reinterpret_cast<const uint4 *>(mine + __builtin_amdgcn_readlane(sel, b));
                acc[b].x -= r.x; acc[b].y -= r.y; acc[b].z -= r.z; acc[b].w -= r.w;
            }
        }
    }
    if (active) {
#pragma unroll
        for (int b = 0; b < T; b++) {
            if (b0 + b >= B) break;
            uint32_t *o = A.out + (size_t)(b0 + b) * (A.n + 1) + 4 * q0;
            const uint32_t v[4] = {acc[b].x, acc[b].y, acc[b].z, acc[b].w};
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (4 * q0 + c <= A.n && v[c]) atomicAdd(o + c, v[c]);
        }
    }
}

// Column-sliced variant for the larger key-switch bases (Uint sets: base 16 ... 128), where a tile of
// whole rows does not fit LDS and T = 32 ciphertexts would use at most half of the base-1 candidate
// rows.  One workgroup = 256 ciphertexts x 64 output columns x IC extracted coefficients.  For each
// (i, j) the base-1 candidate rows' 64-column slice (<= 32 KB) is staged double-buffered in LDS next
// to a zero row; each of the 4 waves walks its own 64 ciphertexts: lane = column, the ciphertext's
// digit is wave-uniform (v_readlane of a per-lane digit word), so the read is one conflict-free
// ds_read_b32 at a scalar-selected row.  Every key row slice crosses L2 once per 256 ciphertexts.
// grid = ceil(B/256) * ceil((n+1)/64) * ranges workgroups (1-D, rounded up to the 8 XCDs); partial sums are
// combined by atomics into the output k_ks_init prepared.  (keyswitch.go:10-37, trlwe_ops.go:10-21)
template <int BB, int CT = 64>
__global__ __launch_bounds__(256) void k_keyswitch_wide(KeySwitchArgs A, int B, int ranges, int ct_tiles, int col_blocks)
{
    // CT = ciphertexts per wave (64 or 128): a workgroup's staged key rows serve 4 CT ciphertexts.  With CT = 128 every key
    // row crosses L2 / the fabric once per 512 ciphertexts instead of once per 256 -- at Uint5 x 512 the two 256-ciphertext
    // tiles of CT = 64 fetch 2.65 GB for a 1.66 GB table (rocprofv3 FETCH_SIZE, profiles/r04_c_pmc_uint5_summary.txt), which is
    // what the kernel's time follows -- and a staged tile and its barrier are amortised over twice the reads.
    static_assert(CT == 64 || CT == 128, "ciphertexts per wave");
    constexpr int base = 1 << BB, C = 64, CQ = C / 4, H = CT / 64;   // one column per lane (two per lane: 256 VGPRs,
                                                                     // one wave per SIMD, 0.82 vs 0.61 ms at Uint5 x 512)
    constexpr int Q = (base - 1) * CQ, R = (Q + 255) / 256;      // staged uint4 per step, per thread
    __shared__ uint32_t rowbuf[2][base][C];                      // [buffer][digit][column]; digit 0 = zeros
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = A.N, t = A.t;
    B = ks_items(A, B);
    // 1-D grid, XCD-aware decode: hardware deals workgroup ids round-robin over the 8 XCDs (each with its own
    // L2), so the ciphertext tiles that share the key rows of one (column block, coefficient range) are placed
    // on the SAME XCD, consecutive in time -- the rows then come from HBM once instead of once per tile
    // (Uint5 x 512: 0.786 -> 0.769 ms; the kernel is VALU-bound, so the gain is small).
    int tile_x, tile_cr;
    {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        tile_x = slot % ct_tiles; tile_cr = (slot / ct_tiles) * 8 + xcd;
    }
    // coefficient range cr of `ranges` (any number, not only a divisor of N: the launcher sizes the grid to fill whole rounds of
    // resident workgroups -- Uint5 x 512: 1,088 workgroups on 768 slots took two rounds, 748 take one)
    if (tile_x * 4 * CT >= B || tile_cr >= col_blocks * ranges) return;      // whole workgroup (the barriers below are workgroup-wide)
    const int cr = tile_cr / col_blocks;
    const int i0 = (int)((long long)cr * N / ranges), IC = (int)((long long)(cr + 1) * N / ranges) - i0;
    const int b0 = tile_x * 4 * CT + w * CT, c0 = (tile_cr % col_blocks) * C;
    const uint32_t prec = 1u << (32 - (1 + BB * t));
    const int wshift = 32 - BB * t;
    if (tid < C) rowbuf[0][0][tid] = rowbuf[1][0][tid] = 0u;
    // this thread's staged quads: q = tid + 256 r -> candidate row k = q/CQ + 1, quad column q%CQ
    const uint4 *src[R];
    uint4 *dst[R];
    bool live[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int q = tid + 256 * r, k1 = q / CQ, qc = q % CQ;
        live[r] = q < Q && c0 + 4 * qc < A.n1p;
        src[r] = reinterpret_cast<const uint4 *>(A.ksk + ((size_t)i0 * t * (base - 1) + k1) * A.n1p + c0) + qc;
        dst[r] = reinterpret_cast<uint4 *>(&rowbuf[0][(q < Q ? k1 : 0) + 1][0]) + qc;
    }
    const size_t pair_stride = (size_t)(base - 1) * A.n1p / 4;   // uint4 per (i, j) pair
    constexpr int buf_quads = base * C / 4;
    auto digit_word = [&](int i, int half) -> uint32_t {         // lane b holds ciphertext b0 + 64 half + b
        const int b = b0 + 64 * half + lane;
        if (b >= B) return 0u;
        const uint32_t *ta = A.trlwe + (size_t)b * 2 * N;
        const uint32_t ai = i == 0 ? ta[0] : ~ta[N - i];                // trlwe_ops.go:13-19
        return (ai + prec) >> wshift;                                     // all t digits, most significant first
    };
    uint4 g[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (live[r]) dst[r][0] = src[r][0];
        else if (tid + 256 * r < Q) dst[r][0] = dst[r][buf_quads] = make_uint4(0, 0, 0, 0);   // columns past the row end
    }
    const int F = IC * t;
#pragma unroll
    for (int r = 0; r < R; r++) g[r] = live[r] ? src[r][(F > 1 ? 1 : 0) * pair_stride] : make_uint4(0, 0, 0, 0);
    uint32_t acc[CT];
#pragma unroll
    for (int b = 0; b < CT; b++) acc[b] = 0u;
    uint32_t wm[H], wnext[H];
#pragma unroll
    for (int hh = 0; hh < H; hh++) { wm[hh] = digit_word(i0, hh); wnext[hh] = IC > 1 ? digit_word(i0 + 1, hh) : 0u; }
    __syncthreads();
    int j = 0, ii = 0;
    for (int f = 0; f < F; f++) {
        const int cur = f & 1, nxt = cur ^ 1;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (live[r]) dst[r][nxt * buf_quads] = g[r];
        {
            const int fn = f + 2 < F ? f + 2 : F - 1;
#pragma unroll
            for (int r = 0; r < R; r++)
                if (live[r]) g[r] = src[r][(size_t)fn * pair_stride];
        }
        // lane b computes, once per step, the byte offset of the row ciphertext b selects; the loop reads it out
        // with one v_readlane per ciphertext (extracting the digit per ciphertext in scalar code instead cost
        // 0.77 vs 0.61 ms at Uint5 x 512)
        const char *tile = reinterpret_cast<const char *>(&rowbuf[cur][0][lane]);
#pragma unroll
        for (int hh = 0; hh < H; hh++) {
            const int sel = (int)((wm[hh] >> (BB * (t - 1 - j))) & (uint32_t)(base - 1)) * (int)(C * sizeof(uint32_t));
#pragma unroll
            for (int b = 0; b < 64; b++)
                acc[64 * hh + b] -= *reinterpret_cast<const uint32_t *>(tile + __builtin_amdgcn_readlane(sel, b));
        }
        __syncthreads();
        if (++j == t) {
            j = 0; ii++;
#pragma unroll
            for (int hh = 0; hh < H; hh++) {
                wm[hh] = wnext[hh];
                wnext[hh] = ii + 1 < IC ? digit_word(i0 + ii + 1, hh) : 0u;
            }
        }
    }
    const int col = c0 + lane;
    if (col <= A.n) {
#pragma unroll
        for (int b = 0; b < CT; b++) {
            if (b0 + b >= B) break;
            if (acc[b]) atomicAdd(A.out + (size_t)(b0 + b) * (A.n + 1) + col, acc[b]);
        }
    }
}

// ------------------------------------------------------------------------------------
// FFT test seams, spectra in the reference FourierPoly layout.
// ------------------------------------------------------------------------------------
static __global__ __launch_bounds__(64) void k_to_fourier(const uint32_t *__restrict__ polys, double *__restrict__ spectra,
                                                    const cd *__restrict__ twt)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    LaneTwiddles tw;
    load_lane_twiddles(tw, twt, lane);
    const uint32_t *poly = polys + (size_t)blockIdx.x * 1024;
    double *fp = spectra + (size_t)blockIdx.x * 1024;
    cd x[8];
#pragma unroll
    for (int a = 0; a < 8; a++)
        x[a] = cd{(double)(int32_t)poly[64 * a + lane], (double)(int32_t)poly[64 * a + lane + 512]};
    fft512_forward(x, sc, twt, tw, lane);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int s = reference_slot_1024(k, lane), base = 8 * (s >> 2) + (s & 3);
        fp[base] = x[k].re;
        fp[base + 4] = x[k].im;
    }
}

static __global__ __launch_bounds__(64) void k_to_poly(const double *__restrict__ spectra, uint32_t *__restrict__ polys,
                                                 const cd *__restrict__ twt)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    LaneTwiddles tw;
    load_lane_twiddles(tw, twt, lane);
    const double *fp = spectra + (size_t)blockIdx.x * 1024;
    uint32_t *poly = polys + (size_t)blockIdx.x * 1024;
    cd x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int s = reference_slot_1024(k, lane), base = 8 * (s >> 2) + (s & 3);
        x[k] = cd{fp[base], fp[base + 4]};
    }
    fft512_inverse(x, sc, twt, tw, lane);
#pragma unroll
    for (int a = 0; a < 8; a++) {
        poly[64 * a + lane] = round_to_torus_wide(x[a].re);
        poly[64 * a + lane + 512] = round_to_torus_wide(x[a].im);
    }
}

// Small helpers for the MUX composition (gates.go:107-114): gather / scatter LWE samples.
static __global__ void k_gather_rows(const uint32_t *__restrict__ src, const int *__restrict__ idx, uint32_t *__restrict__ dst,
                              int n1, const int *__restrict__ count)
{
    const int r = blockIdx.x;
    if (r >= *count) return;
    for (int x = threadIdx.x; x < n1; x += blockDim.x) dst[(size_t)r * n1 + x] = src[(size_t)idx[r] * n1 + x];
}

static __global__ void k_scatter_rows(const uint32_t *__restrict__ src, const int *__restrict__ idx, uint32_t *__restrict__ dst,
                               int n1, const int *__restrict__ count)
{
    const int r = blockIdx.x;
    if (r >= *count) return;
    for (int x = threadIdx.x; x < n1; x += blockDim.x) dst[(size_t)idx[r] * n1 + x] = src[(size_t)r * n1 + x];
}

// ------------------------------------------------------------------------------------
// Seam kernels of the trgsw / trlwe packages (SURVEY.md 8(b) seam 3): element-wise pieces around the external product and the
// key switch, so that trgsw.CMUX, trlwe.SampleExtractIndex and trgsw.IdentityKeySwitching have entry points of their own.
// ------------------------------------------------------------------------------------

// out = a - b  (SUB = true: ct1 - ct0 of CMuxAssign, evaluator.go:93-96) or a + b (ct0 + product, evaluator.go:102-105), mod 2^32
template <bool SUB>
static __global__ void k_torus_addsub(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t *__restrict__ out, size_t words)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) out[i] = SUB ? a[i] - b[i] : a[i] + b[i];
}

// trlwe.SampleExtractIndexAssign for any index k (trlwe_ops.go:10-21): [B][2][N] -> [B][N+1], body last.
static __global__ void k_sample_extract(const uint32_t *__restrict__ trlwe, uint32_t *__restrict__ out, int N, int k)
{
    const uint32_t *A = trlwe + (size_t)blockIdx.x * 2 * N, *Bp = A + N;
    uint32_t *o = out + (size_t)blockIdx.x * (N + 1);
    for (int i = threadIdx.x; i < N; i += blockDim.x) o[i] = i <= k ? A[k - i] : ~A[N + k - i];
    if (threadIdx.x == 0) o[N] = Bp[k];
}

// The TRLWE sample whose extraction at index 0 is a given TLWELv1 (the inverse of the above at k = 0; ~ is an involution, so this is
// exact): lets trgsw.IdentityKeySwitching (keyswitch.go:10-37, input = an extracted sample) run on the fused extract + key-switch
// kernels.  [B][N+1] -> [B][2][N]; only B[0] of the body polynomial is ever read by the key switch, the rest is zeroed.
static __global__ void k_unextract(const uint32_t *__restrict__ lwe1, uint32_t *__restrict__ trlwe, int N)
{
    const uint32_t *p = lwe1 + (size_t)blockIdx.x * (N + 1);
    uint32_t *A = trlwe + (size_t)blockIdx.x * 2 * N, *Bp = A + N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        A[i] = i == 0 ? p[0] : ~p[N - i];
        Bp[i] = i == 0 ? p[N] : 0u;
    }
}

// ------------------------------------------------------------------------------------
// Device-side MUX split (gates.go:107-114): the compact, ascending list of the items whose op code is MUX,
// built without the host ever seeing the op codes.  Three launches: per-block counts, one-block scan, fill.
// ------------------------------------------------------------------------------------
constexpr int kPlanBlock = 1024;                 // items per block (256 threads x 4)

__device__ __forceinline__ int plan_block_scan(int mine, int *sh /* [256] */, int tid)
{
    sh[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? sh[tid - d] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    return sh[tid] - mine;                        // exclusive prefix; sh[255] = block total
}

static __global__ __launch_bounds__(256) void k_mux_count(const uint8_t *__restrict__ ops, int B, int *__restrict__ block_counts,
                                                          int *__restrict__ status)
{
    __shared__ int sh[256];
    const int tid = threadIdx.x, base = blockIdx.x * kPlanBlock + tid * 4;
    int mine = 0, bad = 0;
    for (int k = 0; k < 4; k++)
        if (base + k < B) { const int op = ops[base + k]; mine += op == 10; bad |= op > 10; }
    if (bad) atomicOr(status, kStatusBadOp);
    plan_block_scan(mine, sh, tid);
    if (tid == 0) block_counts[blockIdx.x] = sh[255];
}

// offsets[b] = sum of block_counts[0..b), offsets[nb] = total = *count.
static __global__ __launch_bounds__(256) void k_mux_scan(const int *__restrict__ block_counts, int nb, int *__restrict__ offsets,
                                                         int *__restrict__ count)
{
    __shared__ int sh[256];
    const int tid = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < nb; base += 256) {
        const int mine = base + tid < nb ? block_counts[base + tid] : 0;
        const int ex = plan_block_scan(mine, sh, tid);
        if (base + tid < nb) offsets[base + tid] = carry + ex;
        carry += sh[255];
        __syncthreads();
    }
    if (tid == 0) { offsets[nb] = carry; *count = carry; }
}

static __global__ __launch_bounds__(256) void k_mux_fill(const uint8_t *__restrict__ ops, int B, const int *__restrict__ offsets,
                                                         int *__restrict__ idx)
{
    __shared__ int sh[256];
    const int tid = threadIdx.x, base = blockIdx.x * kPlanBlock + tid * 4;
    int mine = 0;
    for (int k = 0; k < 4; k++) mine += base + k < B && ops[base + k] == 10;
    int at = offsets[blockIdx.x] + plan_block_scan(mine, sh, tid);
    for (int k = 0; k < 4; k++)
        if (base + k < B && ops[base + k] == 10) idx[at++] = base + k;
}

// every item is a MUX: idx = 0..B-1, *count = B
static __global__ void k_mux_all(int B, int *__restrict__ idx, int *__restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) idx[i] = i;
    if (i == 0) *count = B;
}

} // namespace tfhe
