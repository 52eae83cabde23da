/*
 * tfhe_oracle.c -- CPU restatement of go-tfhe's gate-bootstrap hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see tfhe_oracle.h).
 *
 * PARITY UNPINNED by the Go toolchain.  The reference is pure Go, holds no golden vectors for this
 * path, and the image has no Go toolchain: no Go binary has ever produced a vector for it, there is
 * no oracle/_ref, and tests/test_go_golden.py skips (tools/go_golden/README.md: the three commands
 * that turn it green).  What this restatement IS held to: the reference's only numeric known answer
 * and its decrypt-level truth tables (tests/test_oracle_pins.py), the FFT-free exact-integer product
 * below (at N = 1024, L = 3 any correct fp64 FFT must give these words), and -- CORROBORATION, not a
 * pin -- the reference's source text executed under an in-repo interpreter
 * (tools/go_static/gointerp.py: a Go-subset executor written for this repository, with stand-ins for
 * math/cmplx (libc), math/rand (numpy) and goroutines (inlined); tests/golden/goref/,
 * tests/test_goref_vectors.py: bit for bit up to whole bootstraps and every gate at the full 128-bit set).
 *
 * Build with -ffp-contract=off: Go on amd64 never fuses a*b+c, and the reference's
 * complex arithmetic is written as separate multiplies and adds.
 *
 * Citations are file:line under /root/reference.
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ params */

int orc_get_params(int which, orc_params *o)
{
    switch (which) {
    case 0: /* params/params.go:83-112 */
        *o = (orc_params){550, 1024, 10, 3, 6, 2, 7, 5.0e-5, 3.73e-8}; return 0;
    case 1: /* params/params.go:117-146 */
        *o = (orc_params){630, 1024, 10, 3, 6, 2, 8, 3.0517578125e-05, 2.980232238769531e-8}; return 0;
    case 2: /* params/params.go:151-180 */
        *o = (orc_params){700, 1024, 10, 3, 6, 2, 9, 2.0e-5, 2.0e-8}; return 0;
    case 3: /* params/params.go:362-391 */
        *o = (orc_params){1071, 2048, 11, 1, 22, 6, 3, 7.088226765410429399593757e-08,
                          2.2204460492503131e-17}; return 0;
    case 4: /* Uint1, params/params.go:194-232 */
        *o = (orc_params){700, 1024, 10, 2, 10, 2, 8, 2.0e-5, 2.0e-8}; return 0;
    case 5: /* Uint3, params/params.go:277-313 */
        *o = (orc_params){820, 1024, 10, 1, 23, 6, 2, 0.00000251676160959795544987084234,
                          0.00000000000000022204460492503131}; return 0;
    case 6: /* Uint4, params/params.go:318-354 */
        *o = (orc_params){820, 2048, 11, 1, 22, 5, 3, 0.00000251676160959795544987084234,
                          0.00000000000000022204460492503131}; return 0;
    case 7: /* Uint7 / Uint8 share this shape, params/params.go:444-521 */
        *o = (orc_params){1160, 2048, 11, 1, 22, 7, 3, 1.966220007498402695211596e-08,
                          2.2204460492503131e-17}; return 0;
    case 8: /* Uint2, params/params.go:236-265.  The comment there mentions GLWE rank 3 upstream, but
             * the reference's parameter structs carry no rank and the set runs at rank 1 like the rest. */
        *o = (orc_params){687, 512, 9, 1, 18, 4, 3, 0.00002120846893069971872305794214,
                          0.00000000000231841227527049948463}; return 0;
    default: return -1;
    }
}

/* utils/utils.go:11-14 : math.Mod(d,1)*2^32 -> int64 (truncate) -> uint32 (wrap). */
uint32_t orc_f64_to_torus(double d)
{
    double scaled = fmod(d, 1.0) * 4294967296.0;
    return (uint32_t)(int64_t)scaled;
}

/* cloudkey/cloudkey.go:60-71 */
uint32_t orc_decomposition_offset(const orc_params *p)
{
    uint32_t off = 0, half = 1u << (p->Bgbit - 1);
    for (int i = 0; i < p->L; i++)
        off += half * (1u << (32 - (i + 1) * p->Bgbit));
    return off;
}

/* ------------------------------------------------------------------ FFT tables */

struct orc_fft {
    int N, M;           /* ring degree, M = N/2 complex points */
    /* per-evaluator scratch (the reference's evaluator is buffer-pooled / "zero-allocation",
     * evaluator/buffers.go:21-73; one orc_fft per thread plays that role) */
    uint32_t *w_dig, *w_diff, *w_prod;   /* 8N, 2N, 2N words */
    double *w_spec, *w_sa, *w_sb;        /* N doubles each   */
    double *tw_re, *tw_im;     /* forward twiddles, M-1 entries (poly_evaluator.go:114-133) */
    double *twi_re, *twi_im;   /* inverse twiddles, M-1 entries (:135-140) */
};

/* poly_evaluator.go:146-165 -- the in-place swap loop gives the plain bit-reversal
 * permutation for a power-of-two length. */
static void bit_reverse_pairs(double *re, double *im, int len)
{
    int bits = 0;
    while ((1 << bits) < len) bits++;
    for (int i = 0; i < len; i++) {
        int j = 0;
        for (int b = 0; b < bits; b++) if (i & (1 << b)) j |= 1 << (bits - 1 - b);
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
}

orc_fft *orc_fft_new(int N)
{
    orc_fft *f = (orc_fft *)calloc(1, sizeof *f);
    int M = N / 2, H = M / 2;
    f->N = N; f->M = M;
    f->tw_re = (double *)malloc(sizeof(double) * M);
    f->tw_im = (double *)malloc(sizeof(double) * M);
    f->twi_re = (double *)malloc(sizeof(double) * M);
    f->twi_im = (double *)malloc(sizeof(double) * M);
    double *br = (double *)malloc(sizeof(double) * H), *bi = (double *)malloc(sizeof(double) * H);
    double *cr = (double *)malloc(sizeof(double) * H), *ci = (double *)malloc(sizeof(double) * H);
    /* poly_evaluator.go:117-123 : exp(-2 pi i k/M) and its inverse, bit-reversed. */
    for (int k = 0; k < H; k++) {
        double e = -2.0 * M_PI * (double)k / (double)M;
        br[k] = cos(e);  bi[k] = sin(e);
        cr[k] = cos(-e); ci[k] = sin(-e);
    }
    bit_reverse_pairs(br, bi, H);
    bit_reverse_pairs(cr, ci, H);
    /* :128-133 : stage with m groups uses entries 0..m-1 times the fold factor
     * exp(+2 pi i t/(4M)), t = M/(2m). */
    int w = 0;
    for (int m = 1, t = H; m <= H; m <<= 1, t >>= 1) {
        double a = 2.0 * M_PI * (double)t / (double)(4 * M);
        double fr = cos(a), fi = sin(a);
        for (int i = 0; i < m; i++, w++) {
            f->tw_re[w] = br[i] * fr - bi[i] * fi;
            f->tw_im[w] = br[i] * fi + bi[i] * fr;
        }
    }
    /* :135-140 : inverse table, stages in the opposite order. */
    w = 0;
    for (int m = H, t = 1; m >= 1; m >>= 1, t <<= 1) {
        double a = -2.0 * M_PI * (double)t / (double)(4 * M);
        double fr = cos(a), fi = sin(a);
        for (int i = 0; i < m; i++, w++) {
            f->twi_re[w] = cr[i] * fr - ci[i] * fi;
            f->twi_im[w] = cr[i] * fi + ci[i] * fr;
        }
    }
    free(br); free(bi); free(cr); free(ci);
    f->w_dig = (uint32_t *)malloc(sizeof(uint32_t) * 8 * N);
    f->w_diff = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    f->w_prod = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    f->w_spec = (double *)malloc(sizeof(double) * N);
    f->w_sa = (double *)malloc(sizeof(double) * N);
    f->w_sb = (double *)malloc(sizeof(double) * N);
    return f;
}

void orc_fft_twiddles(const orc_fft *f, double *tw_re, double *tw_im, double *twinv_re, double *twinv_im)
{
    size_t bytes = sizeof(double) * (size_t)(f->M - 1);
    memcpy(tw_re, f->tw_re, bytes); memcpy(tw_im, f->tw_im, bytes);
    memcpy(twinv_re, f->twi_re, bytes); memcpy(twinv_im, f->twi_im, bytes);
}

void orc_fft_free(orc_fft *f)
{
    if (!f) return;
    free(f->tw_re); free(f->tw_im); free(f->twi_re); free(f->twi_im);
    free(f->w_dig); free(f->w_diff); free(f->w_prod); free(f->w_spec); free(f->w_sa); free(f->w_sb);
    free(f);
}

/* FourierPoly storage (poly/poly.go:54-62): complex slot c lives at doubles
 * 8*(c/4) + c%4 (re) and +4 (im). */
#define RE(c) (8 * ((c) >> 2) + ((c) & 3))
#define IM(c) (RE(c) + 4)

/* fftInPlace (fourier_transform.go:178-247).  The reference unrolls the stages by
 * block shape (first / middle / second-to-last / last); in complex-slot terms every
 * stage is the same thing: m groups of half-width h = M/(2m), group g pairs slot
 * 2gh+x with 2gh+h+x using twiddle tw[(m-1)+g], butterfly (u,v) -> (u+v*w, u-v*w)
 * (:170-174).  Natural order in, bit-reversed out, no scaling. */
static void fft_forward(const orc_fft *f, double *d)
{
    int M = f->M;
    for (int m = 1; m < M; m <<= 1) {
        int h = M / (2 * m);
        for (int g = 0; g < m; g++) {
            double wr = f->tw_re[m - 1 + g], wi = f->tw_im[m - 1 + g];
            for (int x = 0; x < h; x++) {
                int u = 2 * g * h + x, v = u + h;
                double vr = d[RE(v)], vi = d[IM(v)];
                double pr = vr * wr - vi * wi;
                double pi = vr * wi + vi * wr;
                double ur = d[RE(u)], ui = d[IM(u)];
                d[RE(u)] = ur + pr; d[IM(u)] = ui + pi;
                d[RE(v)] = ur - pr; d[IM(v)] = ui - pi;
            }
        }
    }
}

/* ifftInPlace (fourier_transform.go:258-347): stages m = M/2 ... 1, butterfly
 * (u,v) -> (u+v, (u-v)*w) (:250-255), twInv consumed in that order, and a final
 * division by M fused into the last stage (:315-345). */
static void fft_inverse(const orc_fft *f, double *d)
{
    int M = f->M, w0 = 0;
    for (int m = M / 2; m >= 1; m >>= 1) {
        int h = M / (2 * m);
        for (int g = 0; g < m; g++) {
            double wr = f->twi_re[w0 + g], wi = f->twi_im[w0 + g];
            for (int x = 0; x < h; x++) {
                int u = 2 * g * h + x, v = u + h;
                double ur = d[RE(u)], ui = d[IM(u)], vr = d[RE(v)], vi = d[IM(v)];
                double sr = ur + vr, si = ui + vi, tr = ur - vr, ti = ui - vi;
                d[RE(u)] = sr; d[IM(u)] = si;
                d[RE(v)] = tr * wr - ti * wi;
                d[IM(v)] = tr * wi + ti * wr;
            }
        }
        w0 += m;
    }
    double scale = (double)M;
    for (int i = 0; i < f->N; i++) d[i] /= scale;
}

/* convertPolyToFourierPolyAssign (fourier_transform.go:64-85) then fftInPlace. */
void orc_to_fourier(const orc_fft *f, const uint32_t *p, double *fp)
{
    int M = f->M;
    for (int c = 0; c < M; c++) {
        fp[RE(c)] = (double)(int32_t)p[c];
        fp[IM(c)] = (double)(int32_t)p[c + M];
    }
    fft_forward(f, fp);
}

/* ifftInPlace, floatModQInPlace (:88-104), convertFourierPolyToPolyAssign (:107-125). */
void orc_to_poly(const orc_fft *f, double *fp, uint32_t *p, double *pre_round)
{
    const double Q = 4294967296.0;
    int M = f->M;
    fft_inverse(f, fp);
    if (pre_round) memcpy(pre_round, fp, sizeof(double) * f->N);
    for (int i = 0; i < f->N; i++)
        fp[i] = round(fp[i] - Q * round(fp[i] / Q));   /* Go math.Round = C round() */
    for (int c = 0; c < M; c++) {
        p[c]     = (uint32_t)(int64_t)fp[RE(c)];
        p[c + M] = (uint32_t)(int64_t)fp[IM(c)];
    }
}

/* fourier_ops.go:167-191 : out += v0 * v1, written as out + (ac - bd), out + (ad + bc). */
void orc_fourier_mul_add(int N, const double *v0, const double *v1, double *out)
{
    for (int c = 0; c < N / 2; c++) {
        double a = v0[RE(c)], b = v0[IM(c)], x = v1[RE(c)], y = v1[IM(c)];
        double r = out[RE(c)] + (a * x - b * y);
        double i = out[IM(c)] + (a * y + b * x);
        out[RE(c)] = r; out[IM(c)] = i;
    }
}

/* decomposer.go:55-66 : truncating gadget decomposition, digits stored as u32. */
void orc_decompose(const orc_params *p, const uint32_t *poly, uint32_t offset, uint32_t *out)
{
    uint32_t mask = (1u << p->Bgbit) - 1, half = 1u << (p->Bgbit - 1);
    for (int j = 0; j < p->N; j++) {
        uint32_t tmp = poly[j] + offset;
        for (int l = 0; l < p->L; l++)
            out[l * p->N + j] = ((tmp >> (32 - (l + 1) * p->Bgbit)) & mask) - half;
    }
}

/* buffer_methods.go:133-164.  Written as one rule: with s = (j - k) mod 2N the
 * result coefficient j is a[s] if s < N and (0xFFFFFFFF - a[s-N]) otherwise -- the
 * reference's "negation" is the bitwise complement (:152,158), kept for bit parity. */
void orc_poly_mul_xk(int N, const uint32_t *a, int k, uint32_t *out)
{
    int two = 2 * N;
    k %= two; if (k < 0) k += two;
    for (int j = 0; j < N; j++) {
        int s = j - k; if (s < 0) s += two;
        out[j] = s < N ? a[s] : (0xFFFFFFFFu - a[s - N]);
    }
}

/* Implementation-independent truth: schoolbook negacyclic product mod 2^32. */
void orc_negacyclic_exact(int N, const uint32_t *a, const uint32_t *b, uint32_t *out)
{
    for (int k = 0; k < N; k++) {
        uint32_t acc = 0;
        for (int j = 0; j <= k; j++)       acc += (uint32_t)(int32_t)a[j] * b[k - j];
        for (int j = k + 1; j < N; j++)    acc -= (uint32_t)(int32_t)a[j] * b[N + k - j];
        out[k] = acc;
    }
}

/* ------------------------------------------------------------------ ciphertext path */

/* evaluator.go:50-81 (= trgsw.go:108-134). */
void orc_external_product(const orc_params *p, const orc_fft *f, const double *bsk_i,
                          const uint32_t *in, uint32_t *out)
{
    int N = p->N, L = p->L;
    uint32_t off = orc_decomposition_offset(p);
    uint32_t *dig = f->w_dig;                       /* 2L*N <= 8N words */
    double *spec = f->w_spec, *sa = f->w_sa, *sb = f->w_sb;
    memset(sa, 0, sizeof(double) * N);              /* FourierA/B.Clear() (evaluator.go:69-70) */
    memset(sb, 0, sizeof(double) * N);
    orc_decompose(p, in, off, dig);                 /* A digits -> rows 0..L-1   (:59) */
    orc_decompose(p, in + N, off, dig + L * N);     /* B digits -> rows L..2L-1  (:61) */
    for (int r = 0; r < 2 * L; r++) {               /* same accumulation order as :73-76 */
        orc_to_fourier(f, dig + r * N, spec);
        orc_fourier_mul_add(N, spec, bsk_i + (size_t)(2 * r) * N, sa);
        orc_fourier_mul_add(N, spec, bsk_i + (size_t)(2 * r + 1) * N, sb);
    }
    orc_to_poly(f, sa, out, NULL);
    orc_to_poly(f, sb, out + N, NULL);
}

void orc_external_product_exact(const orc_params *p, const uint32_t *bsk_i, const uint32_t *in,
                                uint32_t *out)
{
    int N = p->N, L = p->L;
    uint32_t off = orc_decomposition_offset(p);
    uint32_t *dig = (uint32_t *)malloc(sizeof(uint32_t) * 2 * L * N);
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * N);
    orc_decompose(p, in, off, dig);
    orc_decompose(p, in + N, off, dig + L * N);
    memset(out, 0, sizeof(uint32_t) * 2 * N);
    for (int r = 0; r < 2 * L; r++)
        for (int part = 0; part < 2; part++) {
            orc_negacyclic_exact(N, dig + r * N, bsk_i + (size_t)(2 * r + part) * N, tmp);
            for (int j = 0; j < N; j++) out[part * N + j] += tmp[j];
        }
    free(dig); free(tmp);
}

/* evaluator.go:85-106 */
void orc_cmux(const orc_params *p, const orc_fft *f, const double *bsk_i, const uint32_t *ct0,
              const uint32_t *ct1, uint32_t *out)
{
    int N2 = 2 * p->N;
    uint32_t *diff = f->w_diff, *prod = f->w_prod;
    for (int j = 0; j < N2; j++) diff[j] = ct1[j] - ct0[j];
    orc_external_product(p, f, bsk_i, diff, prod);
    for (int j = 0; j < N2; j++) out[j] = ct0[j] + prod[j];
}

/* Mod-switch of the body and of a mask word (evaluator.go:116,122). */
static int mod_switch_b(const orc_params *p, uint32_t b)
{
    int64_t v = (int64_t)b + ((int64_t)1 << (31 - p->Nbit - 1));     /* int add: no 32-bit wrap */
    return 2 * p->N - (int)(v >> (32 - p->Nbit - 1));
}
static int mod_switch_a(const orc_params *p, uint32_t a)
{
    return (int)((uint32_t)(a + (1u << (31 - p->Nbit - 1))) >> (32 - p->Nbit - 1)); /* wraps */
}

/* evaluator.go:110-135 */
static void blind_rotate_impl(const orc_params *p, const orc_fft *f, const double *bsk,
                              const uint32_t *bsk_torus, const uint32_t *ct,
                              const uint32_t *tv, int nsteps, uint32_t *out)
{
    int N = p->N, n = p->n;
    size_t stride = (size_t)2 * p->L * 2 * N;
    if (nsteps < 0 || nsteps > n) nsteps = n;
    uint32_t *acc = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    uint32_t *rot = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    uint32_t *prod = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    int bt = mod_switch_b(p, ct[n]);
    orc_poly_mul_xk(N, tv, bt, acc);
    orc_poly_mul_xk(N, tv + N, bt, acc + N);
    for (int i = 0; i < nsteps; i++) {
        int at = mod_switch_a(p, ct[i]);
        orc_poly_mul_xk(N, acc, at, rot);
        orc_poly_mul_xk(N, acc + N, at, rot + N);
        if (f) {
            orc_cmux(p, f, bsk + stride * i, acc, rot, acc);
        } else {
            for (int j = 0; j < 2 * N; j++) rot[j] -= acc[j];
            orc_external_product_exact(p, bsk_torus + stride * i, rot, prod);
            for (int j = 0; j < 2 * N; j++) acc[j] += prod[j];
        }
    }
    memcpy(out, acc, sizeof(uint32_t) * 2 * N);
    free(acc); free(rot); free(prod);
}

void orc_blind_rotate(const orc_params *p, const orc_fft *f, const double *bsk, const uint32_t *ct,
                      const uint32_t *tv, int nsteps, uint32_t *out)
{
    blind_rotate_impl(p, f, bsk, NULL, ct, tv, nsteps, out);
}

void orc_blind_rotate_exact(const orc_params *p, const uint32_t *bsk_torus, const uint32_t *ct,
                            const uint32_t *tv, int nsteps, uint32_t *out)
{
    blind_rotate_impl(p, NULL, NULL, bsk_torus, ct, tv, nsteps, out);
}

/* trlwe_ops.go:10-21 (again the bitwise complement, :17). */
void orc_sample_extract(int N, const uint32_t *trlwe, int k, uint32_t *out)
{
    for (int i = 0; i < N; i++)
        out[i] = i <= k ? trlwe[k - i] : (0xFFFFFFFFu - trlwe[N + k - i]);
    out[N] = trlwe[N + k];
}

/* keyswitch.go:10-37 */
void orc_key_switch(const orc_params *p, const uint32_t *ksk, const uint32_t *lv1, uint32_t *out)
{
    int N = p->N, n = p->n, t = p->t, bb = p->basebit, base = 1 << bb;
    uint32_t prec = 1u << (32 - (1 + bb * t));
    for (int x = 0; x < n; x++) out[x] = 0;
    out[n] = lv1[N];
    for (int i = 0; i < N; i++) {
        uint32_t abar = lv1[i] + prec;
        for (int j = 0; j < t; j++) {
            uint32_t k = (abar >> (32 - (j + 1) * bb)) & (uint32_t)(base - 1);
            if (k) {
                const uint32_t *row = ksk + ((size_t)base * t * i + (size_t)base * j + k) * (n + 1);
                for (int x = 0; x <= n; x++) out[x] -= row[x];
            }
        }
    }
}

/* evaluator.go:139-148 ; programmable_bootstrap.go:93-115 differs only in testvec. */
void orc_bootstrap(const orc_params *p, const orc_fft *f, const double *bsk, const uint32_t *ksk,
                   const uint32_t *ct, const uint32_t *tv, uint32_t *out)
{
    uint32_t *acc = (uint32_t *)malloc(sizeof(uint32_t) * 2 * p->N);
    uint32_t *ext = (uint32_t *)malloc(sizeof(uint32_t) * (p->N + 1));
    orc_blind_rotate(p, f, bsk, ct, tv, -1, acc);
    orc_sample_extract(p->N, acc, 0, ext);
    orc_key_switch(p, ksk, ext, out);
    free(acc); free(ext);
}

/* ------------------------------------------------------------------ gates */

/* gates_helper.go:10-63 (NAND/AND/OR/XOR) and gates.go:52-104 (the rest).
 * out = sa*a + sb*b (all n+1 words) then body += constant.  XNOR follows the tested
 * scalar gate (+1/4, gates.go:56), not BatchXNOR's -1/4 (gates.go:293). */
int orc_gate_prepare(const orc_params *p, int op, const uint32_t *a, const uint32_t *b, uint32_t *out)
{
    uint32_t sa, sb, c;
    const uint32_t E = 0x20000000u /* 1/8 */, Q = 0x40000000u /* 1/4 */;
    switch (op) {
    case ORC_NAND:  sa = -1u; sb = -1u; c = E;   break;
    case ORC_AND:   sa = 1;   sb = 1;   c = -E;  break;
    case ORC_OR:    sa = 1;   sb = 1;   c = E;   break;
    case ORC_XOR:   sa = 1;   sb = 2;   c = Q;   break;
    case ORC_XNOR:  sa = 1;   sb = -2u; c = Q;   break;
    case ORC_NOR:   sa = -1u; sb = -1u; c = -E;  break;
    case ORC_ANDNY: sa = -1u; sb = 1;   c = -E;  break;
    case ORC_ANDYN: sa = 1;   sb = -1u; c = -E;  break;
    case ORC_ORNY:  sa = -1u; sb = 1;   c = E;   break;
    case ORC_ORYN:  sa = 1;   sb = -1u; c = E;   break;
    default: return -1;
    }
    for (int x = 0; x <= p->n; x++) out[x] = sa * a[x] + sb * b[x];
    out[p->n] += c;
    return 0;
}

/* cloudkey.go:74-85 */
void orc_gate_testvec(const orc_params *p, uint32_t *tv)
{
    for (int j = 0; j < p->N; j++) { tv[j] = 0; tv[p->N + j] = 0x20000000u; }
}

int orc_gate(const orc_params *p, const orc_fft *f, const double *bsk, const uint32_t *ksk, int op,
             const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out)
{
    int n1 = p->n + 1, rc = 0;
    uint32_t *tv = (uint32_t *)malloc(sizeof(uint32_t) * 2 * p->N);
    uint32_t *pre = (uint32_t *)malloc(sizeof(uint32_t) * n1);
    orc_gate_testvec(p, tv);
    if (op == ORC_MUX) {            /* gates.go:107-114 : OR(AND(a,b), AND(NOT a, c)) */
        uint32_t *x = (uint32_t *)malloc(sizeof(uint32_t) * n1);
        uint32_t *y = (uint32_t *)malloc(sizeof(uint32_t) * n1);
        uint32_t *na = (uint32_t *)malloc(sizeof(uint32_t) * n1);
        if (!c) rc = -1;
        else {
            orc_gate_prepare(p, ORC_AND, a, b, pre);  orc_bootstrap(p, f, bsk, ksk, pre, tv, x);
            for (int i = 0; i < n1; i++) na[i] = 0u - a[i];          /* NOT = Neg, gates.go:117 */
            orc_gate_prepare(p, ORC_AND, na, c, pre); orc_bootstrap(p, f, bsk, ksk, pre, tv, y);
            orc_gate_prepare(p, ORC_OR, x, y, pre);   orc_bootstrap(p, f, bsk, ksk, pre, tv, out);
        }
        free(x); free(y); free(na);
    } else {
        rc = orc_gate_prepare(p, op, a, b, pre);
        if (!rc) orc_bootstrap(p, f, bsk, ksk, pre, tv, out);
    }
    free(tv); free(pre);
    return rc;
}

/* One bootstrap per worker over independent items (trgsw.go:234-252, gates.go:156-182). */
int orc_gate_batch(const orc_params *p, const double *bsk, const uint32_t *ksk, const uint8_t *ops,
                   int op_uniform, const uint32_t *a, const uint32_t *b, const uint32_t *c,
                   uint32_t *out, int B, int nthreads)
{
    int n1 = p->n + 1, used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        orc_fft *f = orc_fft_new(p->N);
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
        for (int i = 0; i < B; i++) {
            int op = op_uniform >= 0 ? op_uniform : ops[i];
            orc_gate(p, f, bsk, ksk, op, a + (size_t)i * n1, b + (size_t)i * n1,
                     c ? c + (size_t)i * n1 : NULL, out + (size_t)i * n1);
        }
        orc_fft_free(f);
    }
    (void)nthreads;
    return used;
}

int orc_bootstrap_batch(const orc_params *p, const double *bsk, const uint32_t *ksk,
                        const uint32_t *ct, const uint32_t *tv, int tv_per_item, uint32_t *out,
                        int B, int nthreads)
{
    int n1 = p->n + 1, used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        orc_fft *f = orc_fft_new(p->N);
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
        for (int i = 0; i < B; i++)
            orc_bootstrap(p, f, bsk, ksk, ct + (size_t)i * n1,
                          tv + (tv_per_item ? (size_t)i * 2 * p->N : 0), out + (size_t)i * n1);
        orc_fft_free(f);
    }
    (void)nthreads;
    return used;
}
