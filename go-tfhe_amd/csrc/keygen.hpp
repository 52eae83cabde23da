// keygen.hpp -- cloud-key generation on the GPU.
//
// Replaces cloudkey.NewCloudKey (cloudkey/cloudkey.go:24-31) for a caller that holds the secret
// key: genBootstrappingKey (cloudkey.go:123-145 -> TRGSWLv1.EncryptTorus trgsw.go:32-57 ->
// TRLWELv1.EncryptF64 trlwe.go:28-50 -> NewTRGSWLv1FFT trgsw.go:71-82) and genKeySwitchingKey
// (cloudkey.go:88-120 -> TLWELv0.EncryptF64 tlwe.go:36-50), written straight into the engine's
// device layouts, so the 172 MB (1.76 GB at Uint5) host->device key upload disappears.
//
// Randomness: the reference draws from an auto-seeded math/rand (no reproducible seed exists,
// SURVEY.md 3.4); here a counter-based Philox4x32-10 keyed by (seed, stream) gives every sample a
// fixed position, so a (seed, secret key) pair always produces the same cloud key regardless of
// launch geometry.  The seed is SECRET key material: masks and noise of the published cloud key are a
// function of it (whoever knows it can strip the noise and solve for the secret keys), so callers pass 128 bits
// from the OS entropy source (tfhe_keygen_cloud_seeded with seed = NULL does) and fixed seeds only in tests.
// Like the reference's math/rand, Philox is a statistical generator, not a cryptographic one.  Gaussians by Box-Muller in fp64, added on the torus exactly as
// utils.GaussianTorus does (utils/utils.go:31-41: F64ToTorus(normal * stddev)).
#pragma once

#include "kernels.hpp"
#include "kernels_n2048.hpp"
#include "kernels_n512.hpp"

namespace tfhe {

// 128 bits of seed: lo keys the cipher, hi is folded into the two counter words that do not carry the sample
// index (hi = 0 reproduces the 64-bit-seed streams of tfhe_keygen_cloud).
struct Seed128 {
    uint64_t lo, hi;
};

struct Philox {
    uint32_t key[2];
    uint32_t ctr[4];
    __device__ __forceinline__ static void round(uint32_t (&c)[4], uint32_t (&k)[2])
    {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    // 4 random words for (seed, stream, index)
    __device__ __forceinline__ static void block(Seed128 seed, uint32_t stream, uint64_t index, uint32_t (&out)[4])
    {
        uint32_t k[2] = {(uint32_t)seed.lo, (uint32_t)(seed.lo >> 32)};
        uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), stream ^ (uint32_t)seed.hi, 0x7F4E0000u ^ (uint32_t)(seed.hi >> 32)};
#pragma unroll
        for (int r = 0; r < 10; r++) round(c, k);
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

// utils.F64ToTorus (utils/utils.go:11-14) for |d| < 1: trunc(d * 2^32) as int64, wrapped.
__device__ __forceinline__ uint32_t f64_to_torus_small(double d)
{
    return (uint32_t)(long long)(d * 4294967296.0);
}

// One uniform torus word and one Gaussian torus sample (mean 0, stddev alpha) per call.
__device__ __forceinline__ void uniform_and_gaussian(Seed128 seed, uint32_t stream, uint64_t index, double alpha,
                                                     uint32_t &uni, uint32_t &gauss)
{
    uint32_t r[4];
    Philox::block(seed, stream, index, r);
    uni = r[0];
    const double u = ((double)r[1] + 0.5) * (1.0 / 4294967296.0);                  // (0, 1)
    const double v = ((double)r[2] + 0.5) * (1.0 / 4294967296.0);
    const double nrm = sqrt(-2.0 * log(u)) * cospi(2.0 * v);
    gauss = f64_to_torus_small(fmod(nrm * alpha, 1.0));
}

__device__ __forceinline__ uint32_t uniform_word(Seed128 seed, uint32_t stream, uint64_t index)
{
    uint32_t r[4];
    Philox::block(seed, stream, index, r);
    return r[0];
}

constexpr uint32_t kStreamBskA = 1, kStreamKsk = 2;

// ---- bootstrapping key, N = 1024 shapes.  One wave per TRGSW row (i, r): both of its
//      polynomials (A uniform, B = A*s1 + e, gadget term added) leave in Fourier form.
template <int L, int BGBIT>
static __global__ __launch_bounds__(64) void k_keygen_bsk(cd *__restrict__ bsk, const cd *__restrict__ twt,
                                                           const cd *__restrict__ s1_spec /* [8][64] */,
                                                           const uint32_t *__restrict__ s0, double alpha, Seed128 seed)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    const int row = blockIdx.x;                    // i * 2L + r
    const int i = row / (2 * L), r = row % (2 * L);
    const int p = r / L, l = r % L;
    LaneTwiddles tw;
    load_lane_twiddles(tw, twt, lane);
    uint32_t a[16], e[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int j = q < 8 ? 64 * q + lane : 64 * (q - 8) + lane + 512;
        uniform_and_gaussian(seed, kStreamBskA, (uint64_t)row * 1024 + j, alpha, a[q], e[q]);
    }
    cd x[8], as[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = cd{(double)(int32_t)a[k], (double)(int32_t)a[k + 8]};
    fft512_forward(x, sc, twt, tw, lane);                       // spectrum of A
#pragma unroll
    for (int k = 0; k < 8; k++) as[k] = cmul(x[k], s1_spec[k * 64 + lane]);
    fft512_inverse(as, sc, twt, tw, lane);                      // A * s1 (exact: |.| < 2^42)
    uint32_t b[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        b[k] = round_to_torus_small(as[k].re) + e[k];
        b[k + 8] = round_to_torus_small(as[k].im) + e[k + 8];
    }
    // gadget term s0[i] / Bg^(l+1) on coefficient 0 of A (rows < L) or of B (rows >= L)  (trgsw.go:51-54)
    const uint32_t g = s0[i] << (32 - (l + 1) * BGBIT);
    cd y[8];
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = cd{(double)(int32_t)b[k], (double)(int32_t)b[k + 8]};
    if (p == 1 && lane == 0) y[0].re = (double)(int32_t)(b[0] + g);
    fft512_forward(y, sc, twt, tw, lane);                       // spectrum of B
    // Rows < L carry the gadget term on coefficient 0 of A (added AFTER the encryption, so B above was
    // formed from the original A).  A constant term c adds c to every spectral value (Z(w) gains c*w^0);
    // c is the change of the int32 view of coefficient 0, which only lane 0 holds.
    const double a0_fix = p == 0 ? ((double)(int32_t)(a[0] + g) - (double)(int32_t)a[0]) : 0.0;
    const double add = __shfl(a0_fix, 0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        bsk[bsk_index(L, i, p, l, 0, k, lane)] = cd{x[k].re + add, x[k].im};
        bsk[bsk_index(L, i, p, l, 1, k, lane)] = y[k];
    }
}

// Spectrum of the level-1 secret key (binary polynomial), device order [8][64].
static __global__ __launch_bounds__(64) void k_keygen_s1_spectrum(const uint32_t *__restrict__ s1, cd *__restrict__ out,
                                                                   const cd *__restrict__ twt)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    LaneTwiddles tw;
    load_lane_twiddles(tw, twt, lane);
    cd x[8];
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = cd{(double)s1[64 * a + lane], (double)s1[64 * a + lane + 512]};
    fft512_forward(x, sc, twt, tw, lane);
#pragma unroll
    for (int k = 0; k < 8; k++) out[k * 64 + lane] = x[k];
}

// ---- N = 512 (L = 1): one wave per TRGSW sample, half-wave h encrypts row h (0: gadget on A, 1: on B).
//      The secret-key spectrum comes from k_spectra_512 ([8][32]).
template <int BGBIT>
static __global__ __launch_bounds__(64) void k_keygen_bsk_512(cd *__restrict__ bsk, const cd *__restrict__ twt,
                                                               const cd *__restrict__ s1_spec /* [8][32] */,
                                                               const uint32_t *__restrict__ s0, double alpha, Seed128 seed)
{
    __shared__ cd sc[kScratchSlots512];
    const int lane = threadIdx.x, h = lane >> 5, hl = lane & 31;
    const int i = blockIdx.x, row = 2 * i + h;
    cd *sch = sc + h * kHalfScratch;
    LaneTwiddles512 tw;
    load_lane_twiddles_512(tw, twt, hl);
    uint32_t a[16], e[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int j = q < 8 ? 32 * q + hl : 32 * (q - 8) + hl + 256;
        uniform_and_gaussian(seed, kStreamBskA, (uint64_t)row * 512 + j, alpha, a[q], e[q]);
    }
    cd x[8], as[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = cd{(double)(int32_t)a[k], (double)(int32_t)a[k + 8]};
    fft256_forward(x, sch, twt, tw, hl);
#pragma unroll
    for (int k = 0; k < 8; k++) as[k] = cmul(x[k], s1_spec[k * 32 + hl]);
    fft256_inverse(as, sch, twt, tw, hl);                       // A * s1 (exact: |.| < 2^41)
    uint32_t b[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        b[k] = round_to_torus_small(as[k].re) + e[k];
        b[k + 8] = round_to_torus_small(as[k].im) + e[k + 8];
    }
    const uint32_t g = s0[i] << (32 - BGBIT);                   // trgsw.go:51-54 with l = 0
    cd y[8];
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = cd{(double)(int32_t)b[k], (double)(int32_t)b[k + 8]};
    if (h == 1 && hl == 0) y[0].re = (double)(int32_t)(b[0] + g);
    fft256_forward(y, sch, twt, tw, hl);
    const double a0_fix = h == 0 ? ((double)(int32_t)(a[0] + g) - (double)(int32_t)a[0]) : 0.0;
    const double add = __shfl(a0_fix, 32 * h);                  // coefficient 0 lives in lane hl = 0 of the half
#pragma unroll
    for (int k = 0; k < 8; k++) {
        bsk[bsk_index_512(i, h, 0, k, hl)] = cd{x[k].re + add, x[k].im};
        bsk[bsk_index_512(i, h, 1, k, hl)] = y[k];
    }
}

// ---- N = 2048 (L = 1)
template <int BGBIT>
static __global__ __launch_bounds__(64) void k_keygen_bsk_2048(cd *__restrict__ bsk, const cd *__restrict__ twt,
                                                                const cd *__restrict__ s1_spec /* [16][64] */,
                                                                const uint32_t *__restrict__ s0, double alpha, Seed128 seed)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    const int row = blockIdx.x;                    // i * 2 + p
    const int i = row >> 1, p = row & 1;
    LaneTwiddles2048 tw;
    load_lane_twiddles_2048(tw, twt, lane);
    uint32_t a[32], e[32];
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int j = q < 16 ? 64 * q + lane : 64 * (q - 16) + lane + 1024;
        uniform_and_gaussian(seed, kStreamBskA, (uint64_t)row * 2048 + j, alpha, a[q], e[q]);
    }
    cd x[16], as[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = cd{(double)(int32_t)a[k], (double)(int32_t)a[k + 16]};
    fft1024_forward(x, sc, twt, tw, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) as[k] = cmul(x[k], s1_spec[k * 64 + lane]);
    fft1024_inverse(as, sc, twt, tw, lane);
    uint32_t b[32];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        b[k] = round_to_torus_small(as[k].re) + e[k];          // |A*s1| < 2^43: exact
        b[k + 16] = round_to_torus_small(as[k].im) + e[k + 16];
    }
    const uint32_t g = s0[i] << (32 - BGBIT);
    cd y[16];
#pragma unroll
    for (int k = 0; k < 16; k++) y[k] = cd{(double)(int32_t)b[k], (double)(int32_t)b[k + 16]};
    if (p == 1 && lane == 0) y[0].re = (double)(int32_t)(b[0] + g);
    fft1024_forward(y, sc, twt, tw, lane);
    const double a0_fix = p == 0 ? ((double)(int32_t)(a[0] + g) - (double)(int32_t)a[0]) : 0.0;
    const double add = __shfl(a0_fix, 0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        bsk[bsk_index_2048(i, p, 0, k, lane)] = cd{x[k].re + add, x[k].im};
        bsk[bsk_index_2048(i, p, 1, k, lane)] = y[k];
    }
}

static __global__ __launch_bounds__(64) void k_keygen_s1_spectrum_2048(const uint32_t *__restrict__ s1, cd *__restrict__ out,
                                                                        const cd *__restrict__ twt)
{
    __shared__ cd sc[kScratchSlots];
    const int lane = threadIdx.x;
    LaneTwiddles2048 tw;
    load_lane_twiddles_2048(tw, twt, lane);
    cd x[16];
#pragma unroll
    for (int a = 0; a < 16; a++) x[a] = cd{(double)s1[64 * a + lane], (double)s1[64 * a + lane + 1024]};
    fft1024_forward(x, sc, twt, tw, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) out[k * 64 + lane] = x[k];
}

// ---- key-switching key in the packed layout: one wave per packed row (i, j, k >= 1); the last
//      block writes the all-zero padding row.  Row = LWE encryption under s0 of
//      k * s1[i] / 2^((j+1)*basebit)   (cloudkey.go:107-112), an exact torus shift.
static __global__ __launch_bounds__(64) void k_keygen_ksk(uint32_t *__restrict__ ksk, const uint32_t *__restrict__ s0,
                                                           const uint32_t *__restrict__ s1, int n, int n1p, int t, int bb,
                                                           size_t rows_packed, double alpha, Seed128 seed)
{
    const int lane = threadIdx.x;
    const size_t row = blockIdx.x;
    uint32_t *dst = ksk + row * (size_t)n1p;
    if (row + 1 >= rows_packed) {                              // zero padding row
        for (int x = lane; x < n1p; x += 64) dst[x] = 0u;
        return;
    }
    const int base1 = (1 << bb) - 1;
    const size_t ij = row / base1;
    const uint32_t k = (uint32_t)(row % base1) + 1;
    const int i = (int)(ij / t), j = (int)(ij % t);
    uint32_t inner = 0;
    for (int x = lane; x < n1p; x += 64) {
        uint32_t av = 0;
        if (x < n) {
            av = uniform_word(seed, kStreamKsk, row * (uint64_t)(n + 1) + x);
            inner += av * s0[x];
        }
        if (x != n) dst[x] = av;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) inner += __shfl_xor(inner, off);
    if (lane == 0) {
        uint32_t dummy, gs;
        uniform_and_gaussian(seed, kStreamKsk, row * (uint64_t)(n + 1) + n, alpha, dummy, gs);
        const uint32_t mu = (k * s1[i]) << (32 - (j + 1) * bb);
        dst[n] = inner + mu + gs;
    }
}

} // namespace tfhe
