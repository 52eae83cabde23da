"""CPU tier: pin the oracle (oracle/) against everything the reference's own tests hold for
this path, plus the implementation-independent exact-integer product.  No GPU needed."""
import os

import numpy as np
import pytest

from conftest import rand_u32

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# --- the only numeric KAT in the reference: utils/utils_test.go:15-20
@pytest.mark.parametrize("d,want", [(0.0, 0), (0.125, 536870912), (-0.125, 3758096384), (0.25, 1073741824), (0.5, 2147483648)])
def test_f64_to_torus_kat(oracle, d, want):
    assert oracle.f64_to_torus(d) == want


def test_params_and_offset(oracle):
    # params/params_test.go:9-102 (N / n per level) and cloudkey.go:60-71
    # last row: params/uint_params_test.go:228 {"Uint2", SecurityUint2, 512, 687, 4}
    for name, n, N, t in [("80", 550, 1024, 7), ("110", 630, 1024, 8), ("128", 700, 1024, 9), ("uint5", 1071, 2048, 3),
                          ("uint2", 687, 512, 3)]:
        p = oracle.params(name)
        assert (p.n, p.N, p.t) == (n, N, t)
    assert oracle.offset(oracle.params("128")) == 0x82080000
    assert oracle.offset(oracle.params("uint5")) == 0x80000000


@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_fft_round_trip(oracle, N):
    # poly/poly_test.go:10-33 allows |diff| <= 10; the restatement is exact on 32-bit inputs
    rs = np.random.RandomState(N)
    p = rand_u32(rs, N)
    assert np.array_equal(oracle.to_poly(oracle.to_fourier(p)), p)


@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_fft_slot_semantics(oracle, N):
    # slot s holds P(zeta^(1 - 4 bitrev(s))), zeta = exp(i pi / N)  
    rs = np.random.RandomState(7)
    p = rs.randint(-2**20, 2**20, size=N).astype(np.int64)
    fp = oracle.to_fourier(p.astype(np.int32).view(np.uint32)).reshape(-1, 2, 4)
    M, bits = N // 2, int(np.log2(N // 2))
    for s in [0, 1, 2, 5, M // 2 + 3, M - 1]:
        br = int(format(s, f"0{bits}b")[::-1], 2)
        e = (1 - 4 * br) % (2 * N)
        w = np.exp(1j * np.pi * e / N)
        want = np.sum(p * w ** np.arange(N))
        got = fp[s // 4, 0, s % 4] + 1j * fp[s // 4, 1, s % 4]
        assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), s


def test_fft_product_is_exact_integer_product(oracle):
    # N=1024, 6-bit digits x 32-bit key: pre-rounding error ~0.004 << 0.5
    rs = np.random.RandomState(3)
    p = oracle.params("128")
    worst = 0.0
    for _ in range(4):
        dig = rs.randint(-32, 32, size=(6, 1024)).astype(np.int32).view(np.uint32)
        key = rand_u32(rs, (6, 1024))
        acc = np.zeros(1024)
        exact = np.zeros(1024, np.uint32)
        for r in range(6):
            tmp = np.zeros(1024)
            oracle.lib.orc_fourier_mul_add(1024, oracle.to_fourier(dig[r]).ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)),
                                           oracle.to_fourier(key[r]).ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)),
                                           tmp.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)))
            acc += tmp
            exact = (exact + oracle.negacyclic_exact(dig[r], key[r])).astype(np.uint32)
        got, pre = oracle.to_poly(acc, want_pre=True)
        assert np.array_equal(got, exact)
        worst = max(worst, np.abs(pre - np.round(pre)).max())
    assert worst < 0.05, worst


def test_poly_mul_xk_cases(oracle):
    # buffer_methods.go:133-164 incl. the bitwise-complement "negation"
    N = 1024
    a = rand_u32(np.random.RandomState(5), N)
    assert np.array_equal(oracle.poly_mul_xk(a, 0), a)
    assert np.array_equal(oracle.poly_mul_xk(a, 2 * N), a)
    for k in (1, 17, N - 1):
        want = np.concatenate([~a[N - k:], a[:N - k]])
        assert np.array_equal(oracle.poly_mul_xk(a, k), want)
    for k in (N, N + 1, 2 * N - 1):
        kk = k - N
        want = np.concatenate([a[N - kk:], ~a[:N - kk]]) if kk else ~a
        assert np.array_equal(oracle.poly_mul_xk(a, k), want)


def test_decompose_digits(oracle):
    # decomposer.go:55-66: digits in [-Bg/2, Bg/2), recomposition error < 2^(32 - L*Bgbit)
    p = oracle.params("128")
    a = rand_u32(np.random.RandomState(6), p.N)
    dig = oracle.decompose(p, a).view(np.int32).astype(np.int64)
    assert dig.min() >= -32 and dig.max() < 32
    rec = sum(dig[l] << (32 - 6 * (l + 1)) for l in range(3)) % 2**32
    err = (a.astype(np.int64) - rec) % 2**32
    err = np.minimum(err, 2**32 - err)
    assert err.max() < 2**14
    assert not oracle.decompose(p, np.zeros(p.N, np.uint32)).any()     # zero -> all-zero digits (a14 note)


def test_sample_extract_and_keyswitch_restated(oracle, keys_small):
    # independent numpy restatement of trlwe_ops.go:10-21 and keyswitch.go:10-37
    k = keys_small
    p = k.p
    trl = rand_u32(np.random.RandomState(8), (2, p.N))
    ext = oracle.sample_extract(trl)
    want = np.concatenate([[trl[0][0]], ~trl[0][:0:-1], [trl[1][0]]]).astype(np.uint32)
    assert np.array_equal(ext, want)
    out = np.zeros(p.n + 1, np.uint32)
    out[p.n] = ext[p.N]
    prec = np.uint32(1 << (32 - (1 + p.basebit * p.t)))
    for i in range(p.N):
        abar = np.uint32((int(ext[i]) + int(prec)) & 0xFFFFFFFF)
        for j in range(p.t):
            kk = (int(abar) >> (32 - (j + 1) * p.basebit)) & (p.base - 1)
            if kk:
                out = (out - k.ksk[p.base * p.t * i + p.base * j + kk]).astype(np.uint32)
    assert np.array_equal(oracle.key_switch(p, k.ksk, ext), out)


def test_restatement_equals_exact_integer_chain(oracle, keys_small):
    k = keys_small
    ct = rand_u32(np.random.RandomState(9), k.p.n + 1)
    assert np.array_equal(oracle.blind_rotate(k.p, k.bsk, ct, k.tv), oracle.blind_rotate_exact(k.p, k.bsk_torus, ct, k.tv))


TRUTH = {
    "NAND": lambda a, b: not (a and b), "AND": lambda a, b: a and b, "OR": lambda a, b: a or b,
    "XOR": lambda a, b: a != b, "XNOR": lambda a, b: a == b, "NOR": lambda a, b: not (a or b),
    "ANDNY": lambda a, b: (not a) and b, "ANDYN": lambda a, b: a and (not b),
    "ORNY": lambda a, b: (not a) or b, "ORYN": lambda a, b: a or (not b),
}


@pytest.mark.parametrize("op", sorted(TRUTH))
def test_gate_truth_tables_80bit(oracle, keys80, op):
    # gates/gates_test.go:23-366 re-expressed at decrypt level (BASELINE config 1: 80-bit, CPU)
    k = keys80
    A, B = [0, 0, 1, 1], [0, 1, 0, 1]
    a, b = k.enc(A), k.enc(B)
    out, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, op, a, b)
    assert list(k.dec(out)) == [bool(TRUTH[op](bool(x), bool(y))) for x, y in zip(A, B)]


def test_mux_and_not_80bit(oracle, keys80):
    k = keys80
    A = [0, 0, 0, 0, 1, 1, 1, 1]; B = [0, 0, 1, 1, 0, 0, 1, 1]; C = [0, 1, 0, 1, 0, 1, 0, 1]
    a, b, c = k.enc(A), k.enc(B), k.enc(C)
    out, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, "MUX", a, b, c)
    assert list(k.dec(out)) == [bool(y if x else z) for x, y, z in zip(A, B, C)]
    assert list(k.dec((0 - a).astype(np.uint32))) == [not bool(x) for x in A]       # gates.NOT


def test_nand_128bit_single(oracle, keys128):
    k = keys128
    a, b = k.enc([1]), k.enc([1])
    assert not k.dec(oracle.gate(k.p, k.bsk, k.ksk, "NAND", a[0], b[0])[None])[0]


def test_lut_uint5_known_answer(oracle):
    # B[i] = Encode(f((i+32) div 64)) for i < 2016, -Encode(f(0)) above; A = 0
    p = oracle.params("uint5")
    f = [(3 * x + 1) % 32 for x in range(32)]
    tv = oracle.lut_generate(p, f)
    assert not tv[0].any()
    i = np.arange(2048)
    enc = lambda y: (np.asarray(y, np.int64) << 26).astype(np.uint32)          # Encode(y) = y * 2^26
    want = np.where(i < 2016, enc(np.array(f)[((i + 32) // 64) % 32]), (0 - enc(f[0])).astype(np.uint32))
    assert np.array_equal(tv[1], want.astype(np.uint32))
    gold = np.load(os.path.join(GOLDEN, "lut_uint5_identity.npz"))["lut"]
    assert np.array_equal(oracle.lut_generate(p, np.arange(32)), gold)


@pytest.mark.parametrize("fname", ["identity", "complement", "mod16"])
def test_pbs_uint5_decrypt(oracle, fname):
    # params/uint_params_test.go:17-147 (identity / complement / modulo) at Uint5's ring and
    # gadget (N=2048, L=1, Bgbit=22) with a SHORT LWE dimension to keep the KSK small on CPU.
    from conftest import KeySet
    ks = KeySet(oracle, "uint5", 0x7F4E0004, n_override=48, torus=False)
    f = {"identity": lambda x: x, "complement": lambda x: 31 - x, "mod16": lambda x: x % 16}[fname]
    lut = oracle.lut_generate(ks.p, [f(x) for x in range(32)])
    for m in (0, 1, 7, 16, 19, 30, 31):
        ct = oracle.encrypt_message(ks.p, ks.rng, m, 32, ks.s0)
        out = oracle.bootstrap(ks.p, ks.bsk, ks.ksk, ct, lut)
        assert oracle.decrypt_message(ks.p, 32, ks.s0, out) == f(m), (fname, m)


@pytest.mark.parametrize("name,modulus", [("uint1", 2), ("uint2", 4), ("uint3", 8), ("uint4", 16)])
def test_pbs_other_uint_sets_decrypt(oracle, name, modulus):
    # params/uint_params_test.go:24-27: Uint1 (m=2), Uint2 (m=4, N=512), Uint3 (m=8), Uint4 (m=16)
    from conftest import KeySet
    ks = KeySet(oracle, name, 0x7F4E0007, n_override=40, torus=False)
    vals = list(range(modulus)) if modulus <= 8 else [0, 1, 2, modulus // 2, modulus - 3, modulus - 2, modulus - 1]
    for f in (lambda x: x, lambda x: modulus - 1 - x, lambda x: x % (modulus // 2)):
        lut = oracle.lut_generate(ks.p, [f(x) for x in range(modulus)])
        for m in vals:
            ct = oracle.encrypt_message(ks.p, ks.rng, m, modulus, ks.s0)
            out = oracle.bootstrap(ks.p, ks.bsk, ks.ksk, ct, lut)
            assert oracle.decrypt_message(ks.p, modulus, ks.s0, out) == f(m), (name, m)


def test_pbs_binary_80bit(oracle, keys80):
    # evaluator/programmable_bootstrap_test.go:13-188: identity / NOT / constant with m = 2
    k = keys80
    for table, want in [([0, 1], lambda x: x), ([1, 0], lambda x: 1 - x), ([1, 1], lambda x: 1)]:
        lut = oracle.lut_generate(k.p, table)
        for m in (0, 1):
            ct = oracle.encrypt_message(k.p, k.rng, m, 2, k.s0)
            out = oracle.bootstrap(k.p, k.bsk, k.ksk, ct, lut)
            assert oracle.decrypt_message(k.p, 2, k.s0, out) == want(m)


def test_golden_external_product(oracle):
    g = np.load(os.path.join(GOLDEN, "extprod_N1024_L3_Bg6.npz"))
    p = oracle.params("128")
    assert np.array_equal(oracle.external_product_exact(p, g["trgsw_torus"], g["trlwe_in"]), g["trlwe_out"])
    bf = np.stack([oracle.to_fourier(x) for x in g["trgsw_torus"].reshape(-1, 1024)]).reshape(6, 2, 1024)
    assert np.array_equal(oracle.external_product(p, bf, g["trlwe_in"]), g["trlwe_out"])


def test_golden_bootstrap_chain(oracle):
    from conftest import KeySet
    g = np.load(os.path.join(GOLDEN, "bootstrap_n6_seed7F4E0011.npz"))
    ks = KeySet(oracle, "128", int(g["seed"]), n_override=6)
    for i, ct in enumerate(g["lwe_in"]):
        acc = oracle.blind_rotate(ks.p, ks.bsk, ct, ks.tv)
        assert np.array_equal(acc, g["trlwe_acc"][i])
        assert np.array_equal(oracle.key_switch(ks.p, ks.ksk, oracle.sample_extract(acc)), g["lwe_out"][i])


# ---- the oracle's one restructuring, held to the reference's own loop structure --------------------
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_generic_fft_loop_equals_the_four_stage_shapes_bitwise(oracle, N):
    """oracle/tfhe_oracle.c runs fftInPlace / ifftInPlace as one generic loop over complex slots; the reference writes four
    (three) stage shapes over [4 re | 4 im] blocks (fourier_transform.go:178-347).  tests/go_fft_shapes.py keeps the
    reference's structure; on the SAME twiddle table both must produce the same doubles bit for bit -- the spectra a Go shim
    would upload through tfhe_load_bsk_fourier are in exactly this slot order.  (Pins the oracle's restructuring; says nothing
    about the Go binary's own bits -- parity stays 'unpinned')"""
    import go_fft_shapes as g
    tw, tw_inv = oracle.fft_twiddles(N)
    tw, tw_inv = [complex(x) for x in tw], [complex(x) for x in tw_inv]
    rs = np.random.RandomState(N)
    p = rs.randint(0, 2**32, size=N, dtype=np.uint64).astype(np.uint32)
    want = oracle.to_fourier(p)
    c = g.fold(p)
    g.fft_in_place(c, tw)
    got = np.array(c)
    assert got.tobytes() == want.tobytes(), f"forward: {np.count_nonzero(got != want)} of {N} doubles differ"
    # inverse on a spectrum of realistic magnitude (a product of two transforms' size), pre-rounding doubles compared
    spec = want * 1024.0 + rs.standard_normal(N) * 2.0**40
    _, pre = oracle.to_poly(spec, want_pre=True)
    c = [float(x) for x in spec]
    g.ifft_in_place(c, tw_inv)
    got = np.array(c)
    assert got.tobytes() == pre.tobytes(), f"inverse: {np.count_nonzero(got != pre)} of {N} doubles differ"


@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_twiddle_table_construction_follows_the_reference(oracle, N):
    """genTwiddleFactors (poly_evaluator.go:114-143) restated with the reference's in-place bit reversal and fold factors:
    same entries in the same order as the oracle's table.  cos / sin come from different libraries (Go's math.Sincos is a pure-Go
    Cephes port, the oracle uses glibc, this test Python's cmath), so entries may differ in the last bit: <= 2 ulp, not bitwise."""
    import go_fft_shapes as g
    tw, tw_inv = oracle.fft_twiddles(N)
    gtw, gtw_inv = g.gen_twiddle_factors(N // 2)
    assert len(gtw) == len(tw) == N // 2 - 1 and len(gtw_inv) == len(tw_inv)
    assert np.max(np.abs(np.array(gtw) - tw)) <= 4 * np.finfo(np.float64).eps
    assert np.max(np.abs(np.array(gtw_inv) - tw_inv)) <= 4 * np.finfo(np.float64).eps
