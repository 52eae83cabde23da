/* The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no race detector or
 * sanitizer run; the restatement that every parity claim rests on gets one).  Built and run by tests/test_oracle_sanitizers.py with
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off -fopenmp oracle/*.c tests/cpp/oracle_sanitize.c -lm
 * It walks every entry point of oracle/tfhe_oracle.h at reduced LWE dimensions -- key generation, encryption, all eleven gates (batch,
 * threaded), the exact-integer chain, programmable bootstraps through generated tables at three ring shapes -- and checks decryptions, so
 * that the sanitizers see the real access patterns (ragged strides of n + 1 words, the key-switch table's k = 0 rows, the [4 re | 4 im]
 * blocks).  Unsigned wrap-around is the arithmetic of the torus and is not flagged by -fsanitize=undefined (it is defined behaviour).
 * Test infrastructure: nothing here is linked into the product. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/tfhe_oracle.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)

static void one_set(int which, int n, int modulus)
{
    orc_params p;
    CHECK(orc_get_params(which, &p) == 0);
    p.n = n;
    const int N = p.N, n1 = p.n + 1, base = 1 << p.basebit;
    orc_rng r;
    orc_rng_seed(&r, 0x5A17u + (uint64_t)which);
    uint32_t *s0 = malloc(sizeof(uint32_t) * p.n), *s1 = malloc(sizeof(uint32_t) * N);
    orc_keygen_secret(&p, &r, s0, s1);
    const size_t trgsw = (size_t)2 * p.L * 2 * N;
    uint32_t *bsk_t = malloc(sizeof(uint32_t) * trgsw * p.n);
    double *bsk = malloc(sizeof(double) * trgsw * p.n);
    orc_keygen_bsk(&p, &r, s0, s1, bsk_t, bsk);
    uint32_t *ksk = malloc(sizeof(uint32_t) * (size_t)N * p.t * base * n1);
    orc_keygen_ksk(&p, &r, s0, s1, ksk);
    orc_fft *f = orc_fft_new(N);
    uint32_t *tv = malloc(sizeof(uint32_t) * 2 * N);

    if (modulus == 0) {                                  /* a gate set: truth tables through the batch entry point, two threads */
        enum { B = 8 };
        uint32_t *a = malloc(sizeof(uint32_t) * B * n1), *b = malloc(sizeof(uint32_t) * B * n1), *c = malloc(sizeof(uint32_t) * B * n1),
                 *out = malloc(sizeof(uint32_t) * B * n1);
        for (int i = 0; i < B; i++) {
            orc_tlwe_encrypt_bool(&p, &r, i & 1, s0, a + (size_t)i * n1);
            orc_tlwe_encrypt_bool(&p, &r, (i >> 1) & 1, s0, b + (size_t)i * n1);
            orc_tlwe_encrypt_bool(&p, &r, (i >> 2) & 1, s0, c + (size_t)i * n1);
        }
        for (int op = 0; op <= ORC_MUX; op++) {
            CHECK(orc_gate_batch(&p, bsk, ksk, NULL, op, a, b, op == ORC_MUX ? c : NULL, out, B, 2) >= 1);
            for (int i = 0; i < B; i++) {
                const int x = i & 1, y = (i >> 1) & 1, z = (i >> 2) & 1;
                const int want[] = {!(x && y), x && y, x || y, x != y, x == y, !(x || y), !x && y, x && !y, !x || y, x || !y, x ? y : z};
                CHECK(orc_tlwe_decrypt_bool(&p, s0, out + (size_t)i * n1) == want[op]);
            }
        }
        uint8_t ops[B];
        for (int i = 0; i < B; i++) ops[i] = (uint8_t)((i * 5) % 11);
        CHECK(orc_gate_batch(&p, bsk, ksk, ops, -1, a, b, c, out, B, 0) >= 1);
        /* the fp64 chain against exact integers (whole blind rotate), then extract + key switch */
        uint32_t *acc = malloc(sizeof(uint32_t) * 2 * N), *acx = malloc(sizeof(uint32_t) * 2 * N), *ext = malloc(sizeof(uint32_t) * (N + 1)),
                 *lwe = malloc(sizeof(uint32_t) * n1);
        orc_gate_testvec(&p, tv);
        orc_blind_rotate(&p, f, bsk, a, tv, -1, acc);
        orc_blind_rotate_exact(&p, bsk_t, a, tv, -1, acx);
        CHECK(memcmp(acc, acx, sizeof(uint32_t) * 2 * N) == 0);
        orc_sample_extract(N, acc, 0, ext);
        orc_key_switch(&p, ksk, ext, lwe);
        orc_bootstrap(&p, f, bsk, ksk, a, tv, out);
        CHECK(memcmp(lwe, out, sizeof(uint32_t) * n1) == 0);
        uint32_t *prod = malloc(sizeof(uint32_t) * N);
        orc_negacyclic_exact(N, acc, acc + N, prod);
        orc_poly_mul_xk(N, acc, 2 * N - 1, prod);
        orc_poly_mul_xk(N, acc, 0, prod);
        free(prod); free(acc); free(acx); free(ext); free(lwe); free(a); free(b); free(c); free(out);
    } else {                                             /* a Uint set: programmable bootstraps through a generated table */
        int32_t *table = malloc(sizeof(int32_t) * modulus);
        for (int x = 0; x < modulus; x++) table[x] = (3 * x + 1) % modulus;
        orc_lut_generate(&p, table, modulus, tv);
        enum { B = 5 };
        uint32_t *in = malloc(sizeof(uint32_t) * B * n1), *out = malloc(sizeof(uint32_t) * B * n1);
        for (int i = 0; i < B; i++) orc_tlwe_encrypt_message(&p, &r, (i * 7) % modulus, modulus, s0, in + (size_t)i * n1);
        CHECK(orc_bootstrap_batch(&p, bsk, ksk, in, tv, 0, out, B, 2) >= 1);
        for (int i = 0; i < B; i++) CHECK(orc_tlwe_decrypt_message(&p, modulus, s0, out + (size_t)i * n1) == table[(i * 7) % modulus]);
        free(table); free(in); free(out);
    }
    /* the transform's own entry points */
    double *fp = malloc(sizeof(double) * N), *pre = malloc(sizeof(double) * N);
    uint32_t *poly = malloc(sizeof(uint32_t) * N), *back = malloc(sizeof(uint32_t) * N), *dec = malloc(sizeof(uint32_t) * p.L * N);
    for (int j = 0; j < N; j++) poly[j] = (uint32_t)orc_rng_u64(&r);
    orc_to_fourier(f, poly, fp);
    orc_to_poly(f, fp, back, pre);
    CHECK(memcmp(poly, back, sizeof(uint32_t) * N) == 0);
    orc_decompose(&p, poly, orc_decomposition_offset(&p), dec);
    double *twr = malloc(sizeof(double) * N), *twi = malloc(sizeof(double) * N), *ivr = malloc(sizeof(double) * N), *ivi = malloc(sizeof(double) * N);
    orc_fft_twiddles(f, twr, twi, ivr, ivi);
    free(twr); free(twi); free(ivr); free(ivi);
    free(fp); free(pre); free(poly); free(back); free(dec);
    orc_fft_free(f);
    free(tv); free(ksk); free(bsk); free(bsk_t); free(s0); free(s1);
}

int main(void)
{
    one_set(2, 5, 0);        /* 128-bit ring (N = 1024, L = 3), basebit 2, t = 9 */
    one_set(0, 3, 0);        /* 80-bit: t = 7 */
    one_set(3, 3, 32);       /* Uint5: N = 2048, L = 1, basebit 6 */
    one_set(8, 3, 4);        /* Uint2: N = 512 */
    one_set(4, 3, 2);        /* Uint1: L = 2 */
    CHECK(orc_f64_to_torus(0.125) == 0x20000000u && orc_f64_to_torus(-0.125) == 0xE0000000u);
    if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    puts("oracle under ASan + UBSan: ok");
    return 0;
}
