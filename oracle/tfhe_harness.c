/*
 * tfhe_harness.c -- seeded key generation / encryption / decryption / LUT generation.
 *
 * TEST INFRASTRUCTURE ONLY (see tfhe_oracle.h).  These restate the reference's
 * setup code so that tests and bench.py can make VALID keys and ciphertexts; the
 * reference draws from an auto-seeded math/rand (key/key.go:17, tlwe/tlwe.go:37,
 * trlwe/trlwe.go:29), so no reference seed exists to reproduce -- the PRNG here
 * (xoshiro256** + Box-Muller) is our own and is documented as such.
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ PRNG */

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_rng_seed(orc_rng *r, uint64_t seed)
{
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed);
    r->have_spare = 0; r->spare = 0.0;
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

uint64_t orc_rng_u64(orc_rng *r)
{
    uint64_t *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}

static uint32_t rng_u32(orc_rng *r) { return (uint32_t)(orc_rng_u64(r) >> 32); }
static double rng_unit(orc_rng *r) { return ((double)(orc_rng_u64(r) >> 11) + 0.5) / 9007199254740992.0; }

double orc_rng_normal(orc_rng *r)
{
    if (r->have_spare) { r->have_spare = 0; return r->spare; }
    double u = rng_unit(r), v = rng_unit(r);
    double m = sqrt(-2.0 * log(u)), a = 2.0 * M_PI * v;
    r->spare = m * sin(a); r->have_spare = 1;
    return m * cos(a);
}

/* utils/utils.go:31-41 : F64ToTorus(mu) + F64ToTorus(normal*stddev). */
static uint32_t gaussian_torus(orc_rng *r, double mu, double stddev)
{
    return orc_f64_to_torus(mu) + orc_f64_to_torus(orc_rng_normal(r) * stddev);
}

/* ------------------------------------------------------------------ keys and LWE */

/* key/key.go:16-45 : uniform binary keys. */
void orc_keygen_secret(const orc_params *p, orc_rng *r, uint32_t *s0, uint32_t *s1)
{
    for (int i = 0; i < p->n; i++) s0[i] = (uint32_t)(orc_rng_u64(r) >> 63);
    for (int i = 0; i < p->N; i++) s1[i] = (uint32_t)(orc_rng_u64(r) >> 63);
}

/* tlwe/tlwe.go:36-50 : uniform mask, body = <a,s> + gaussian(mu). */
void orc_tlwe_encrypt_f64(const orc_params *p, orc_rng *r, double mu, double alpha,
                          const uint32_t *s0, uint32_t *ct)
{
    uint32_t inner = 0;
    for (int i = 0; i < p->n; i++) { ct[i] = rng_u32(r); inner += s0[i] * ct[i]; }
    ct[p->n] = inner + gaussian_torus(r, mu, alpha);
}

/* tlwe/tlwe.go:53-61 */
void orc_tlwe_encrypt_bool(const orc_params *p, orc_rng *r, int bit, const uint32_t *s0, uint32_t *ct)
{
    orc_tlwe_encrypt_f64(p, r, bit ? 0.125 : -0.125, p->alpha_lv0, s0, ct);
}

uint32_t orc_tlwe_phase(const orc_params *p, const uint32_t *s0, const uint32_t *ct)
{
    uint32_t inner = 0;
    for (int i = 0; i < p->n; i++) inner += ct[i] * s0[i];
    return ct[p->n] - inner;
}

/* tlwe/tlwe.go:64-73 */
int orc_tlwe_decrypt_bool(const orc_params *p, const uint32_t *s0, const uint32_t *ct)
{
    return (int32_t)orc_tlwe_phase(p, s0, ct) >= 0;
}

/* tlwe/programmable_encrypt.go:12-26 : message * (2^31/modulus) / 2^32. */
void orc_tlwe_encrypt_message(const orc_params *p, orc_rng *r, int msg, int modulus,
                              const uint32_t *s0, uint32_t *ct)
{
    double scale = 2147483648.0 / (double)modulus;
    msg %= modulus; if (msg < 0) msg += modulus;
    orc_tlwe_encrypt_f64(p, r, (double)msg * scale / 4294967296.0, p->alpha_lv0, s0, ct);
}

/* tlwe/programmable_encrypt.go:32-54 */
int orc_tlwe_decrypt_message(const orc_params *p, int modulus, const uint32_t *s0, const uint32_t *ct)
{
    uint32_t scale = (uint32_t)(2147483648u / (uint32_t)modulus);
    uint32_t phase = orc_tlwe_phase(p, s0, ct);
    int decoded = (int)((uint32_t)(phase + scale / 2) / scale);
    return decoded % modulus;
}

/* ------------------------------------------------------------------ cloud key */

/* trlwe/trlwe.go:28-50 with an all-zero plaintext: A uniform, B = gaussian(0) + A*s1.
 * The reference forms A*s1 with its FFT MulPoly (poly_mul.go:12-22); for a binary key
 * the products stay below 2^42, so the FFT result equals this exact integer product. */
static void trlwe_encrypt_zero(const orc_params *p, orc_rng *r, const uint32_t *s1, uint32_t *ab)
{
    int N = p->N;
    uint32_t *as = (uint32_t *)malloc(sizeof(uint32_t) * N);
    for (int j = 0; j < N; j++) ab[j] = rng_u32(r);
    for (int j = 0; j < N; j++) ab[N + j] = gaussian_torus(r, 0.0, p->alpha_lv1);
    orc_negacyclic_exact(N, s1, ab, as);     /* s1 in {0,1}: int32 view is exact */
    for (int j = 0; j < N; j++) ab[N + j] += as[j];
    free(as);
}

/* cloudkey.go:123-145 -> trgsw.go:32-57 (EncryptTorus) -> trgsw.go:71-82 (to Fourier). */
void orc_keygen_bsk(const orc_params *p, orc_rng *r, const uint32_t *s0, const uint32_t *s1,
                    uint32_t *bsk_torus, double *bsk_fourier)
{
    int N = p->N, L = p->L;
    size_t row = (size_t)2 * N, stride = row * 2 * L;
    uint32_t *g = (uint32_t *)malloc(sizeof(uint32_t) * L);
    uint64_t *sub = (uint64_t *)malloc(sizeof(uint64_t) * p->n);
    for (int l = 0; l < L; l++)                       /* trgsw.go:38-42 : 1/Bg^(l+1) */
        g[l] = orc_f64_to_torus(1.0 / pow((double)(1u << p->Bgbit), (double)(l + 1)));
    /* one sub-stream per key element so the n encryptions can run in parallel
     * (the reference also encrypts them concurrently, cloudkey.go:127-142) */
    for (int i = 0; i < p->n; i++) sub[i] = orc_rng_u64(r);
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        orc_fft *f = bsk_fourier ? orc_fft_new(N) : NULL;
        uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * stride);
        orc_rng ri;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int i = 0; i < p->n; i++) {
            uint32_t *dst = bsk_torus ? bsk_torus + stride * i : tmp;
            orc_rng_seed(&ri, sub[i]);
            for (int rr = 0; rr < 2 * L; rr++) trlwe_encrypt_zero(p, &ri, s1, dst + row * rr);
            for (int l = 0; l < L; l++) {                 /* trgsw.go:51-54 */
                dst[row * l] += s0[i] * g[l];             /* rows l   : A[0] */
                dst[row * (l + L) + N] += s0[i] * g[l];   /* rows l+L : B[0] */
            }
            if (bsk_fourier)
                for (int q = 0; q < 4 * L; q++)
                    orc_to_fourier(f, dst + (size_t)q * N, bsk_fourier + stride * i + (size_t)q * N);
        }
        free(tmp); orc_fft_free(f);
    }
    free(g); free(sub);
}

/* cloudkey.go:88-120 : row base*t*i + base*j + k encrypts k*s1[i]/2^((j+1)*basebit)
 * under s0 with the level-0 noise; k = 0 rows stay zero. */
void orc_keygen_ksk(const orc_params *p, orc_rng *r, const uint32_t *s0, const uint32_t *s1, uint32_t *ksk)
{
    int base = 1 << p->basebit, n1 = p->n + 1;
    memset(ksk, 0, sizeof(uint32_t) * (size_t)p->N * p->t * base * n1);
    for (int i = 0; i < p->N; i++)
        for (int j = 0; j < p->t; j++)
            for (int k = 1; k < base; k++) {
                double mu = ((double)k * (double)s1[i]) / (double)((uint64_t)1 << ((j + 1) * p->basebit));
                size_t idx = (size_t)base * p->t * i + (size_t)base * j + k;
                orc_tlwe_encrypt_f64(p, r, mu, p->alpha_lv0, s0, ksk + idx * n1);
            }
}

/* ------------------------------------------------------------------ LUT */

static int div_round(int a, int b) { return (a + b / 2) / b; }   /* lut/generator.go:171-173 */

/* lut/generator.go:56-100 with Encode from lut/encoder.go:17-30,47-74 (scale 1/(2m)). */
void orc_lut_generate(const orc_params *p, const int32_t *table, int modulus, uint32_t *tv)
{
    int N = p->N;
    uint32_t *raw = (uint32_t *)calloc(N, sizeof(uint32_t));
    for (int x = 0; x < modulus; x++) {
        int start = div_round(x * N, modulus), end = div_round((x + 1) * N, modulus);
        int y = table[x] % modulus; if (y < 0) y += modulus;
        uint32_t enc = orc_f64_to_torus((double)y * (1.0 / (double)(2 * modulus)));
        for (int xx = start; xx < end; xx++) raw[xx] = enc;
    }
    int offset = div_round(N, 2 * modulus);
    for (int i = 0; i < N; i++) {
        uint32_t v = raw[(i + offset) % N];
        tv[i] = 0;
        tv[N + i] = i >= N - offset ? 0u - v : v;
    }
    free(raw);
}
