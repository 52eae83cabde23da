"""GPU parity at Uint2 (n=687, N=512, L=1, Bgbit=18, basebit=4, t=3; params.go:236-265): the half-wave
256-point transform and the one-wave-per-bootstrap blind rotate of csrc/kernels_n512.hpp.

Tolerance regime like the other Uint sets (digits up to 2^17 times 2^31 key words summed over 1024
terms reach ~2^58 > 2^53).  Stated tolerances: one external product within 2^9 torus ulps of the exact
integer result per coefficient (the oracle's own deviation is asserted against the same bound); end to
end, identical DecryptLWEMessage and output phase within 2^32/(4*4) of the ideal encoding.  Sample
extract + key switch (base 16) stay bit-exact.  The full-dimension run with an on-GPU generated cloud
key is test_gpu_uint5.py::test_pbs_other_uint_sets[uint2-4]."""
import numpy as np
import pytest

from conftest import KeySet, gpu_params, rand_u32

pytestmark = pytest.mark.gpu
N = 512


def circ_dist(a, b):
    d = (a.astype(np.int64) - b.astype(np.int64)) % 2**32
    return np.minimum(d, 2**32 - d)


@pytest.fixture(scope="module")
def keys_u2(oracle):
    return KeySet(oracle, "uint2", 0x7F4E0010, n_override=48)


@pytest.fixture(scope="module")
def ck_u2(pkg, keys_u2):
    k = keys_u2
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    yield ck
    ck.close()


def test_fft_512_layout_and_round_trip(oracle, ck_u2):
    rs = np.random.RandomState(41)
    polys = rand_u32(rs, (5, N))                          # odd count: the last wave has one idle half
    polys[0] = 0; polys[0][1] = 1                         # X -> the evaluation points themselves
    got = ck_u2.ctx.to_fourier_batch(polys)
    for k in range(5):
        want = oracle.to_fourier(polys[k])
        assert np.abs(got[k] - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), k
    assert np.array_equal(ck_u2.ctx.to_poly_batch(got), polys)
    spectra = np.stack([oracle.to_fourier(p) for p in polys])
    assert np.array_equal(ck_u2.ctx.to_poly_batch(spectra), polys)
    one = ck_u2.ctx.to_fourier_batch(polys[3:4])
    assert np.array_equal(one[0], got[3])


def test_external_product_within_tolerance(oracle, keys_u2, ck_u2):
    k = keys_u2
    trl = rand_u32(np.random.RandomState(42), (4, 2, N))
    worst = 0
    for idx in (0, 17, k.p.n - 1):
        got = ck_u2.ctx.external_product_batch(idx, trl)
        for b in range(4):
            exact = oracle.external_product_exact(k.p, k.bsk_torus[idx], trl[b])
            ref = oracle.external_product(k.p, k.bsk[idx], trl[b])
            worst = max(worst, int(circ_dist(got[b], exact).max()))
            assert circ_dist(got[b], exact).max() <= 2**9, (idx, b, circ_dist(got[b], exact).max())
            assert circ_dist(ref, exact).max() <= 2**9
    print("uint2 external product: worst |GPU - exact| =", worst, "ulps")
    z = ck_u2.ctx.external_product_batch(0, np.zeros((1, 2, N), np.uint32))
    assert not z.any()


def test_bsk_torus_upload_matches_fourier_upload(pkg, keys_u2, ck_u2):
    k = keys_u2
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_torus=k.bsk_torus, ksk=k.ksk)
    trl = rand_u32(np.random.RandomState(43), (2, 2, N))
    assert circ_dist(ck.ctx.external_product_batch(3, trl), ck_u2.ctx.external_product_batch(3, trl)).max() <= 2**9
    ck.close()


def test_extract_keyswitch_bit_exact(oracle, keys_u2, ck_u2):
    k = keys_u2
    trl = rand_u32(np.random.RandomState(44), (3, 2, N))
    got = ck_u2.ctx.extract_keyswitch_batch(trl)
    for b in range(3):
        assert np.array_equal(got[b], oracle.key_switch(k.p, k.ksk, oracle.sample_extract(trl[b])))


def test_single_cmux_step_within_tolerance(oracle, keys_u2, ck_u2):
    # one step of the chain: masks are still comparable (no digit has been re-decomposed yet)
    k = keys_u2
    rs = np.random.RandomState(45)
    cts = rand_u32(rs, (3, k.p.n + 1))
    tv = rand_u32(rs, (2, N))
    got = ck_u2.ctx.blind_rotate_batch(cts, tv, nsteps=1)
    zero = ck_u2.ctx.blind_rotate_batch(cts, tv, nsteps=0)
    for b in range(3):
        assert np.array_equal(zero[b].reshape(-1), oracle.blind_rotate(k.p, k.bsk, cts[b], tv, 0).reshape(-1))
        ref = oracle.blind_rotate(k.p, k.bsk, cts[b], tv, 1)
        assert circ_dist(got[b].reshape(-1), ref.reshape(-1)).max() <= 2**10


FUNCS = {"identity": lambda x: x, "complement": lambda x: 3 - x, "mod2": lambda x: x % 2}


@pytest.mark.parametrize("fname", sorted(FUNCS))
def test_pbs_small_n(oracle, pkg, keys_u2, ck_u2, fname):
    # params/uint_params_test.go:17-147 at messageModulus 4
    k, f = keys_u2, FUNCS[fname]
    msgs = [0, 1, 2, 3, 3, 0]
    lut = oracle.lut_generate(k.p, [f(x) for x in range(4)])
    cts = np.stack([oracle.encrypt_message(k.p, k.rng, m, 4, k.s0) for m in msgs])
    out = ck_u2.ctx.bootstrap_batch(cts, lut)
    assert [oracle.decrypt_message(k.p, 4, k.s0, np.ascontiguousarray(o)) for o in out] == [f(m) for m in msgs]
    for o, m in zip(out, msgs):
        ph = oracle.phase(k.p, k.s0, np.ascontiguousarray(o))
        ideal = np.array([(f(m) << 29) & 0xFFFFFFFF], np.uint32)        # message / (2*modulus)
        assert circ_dist(np.array([ph], np.uint32), ideal)[0] < 2**28
    ev = pkg.evaluator.Evaluator(ck_u2)
    assert oracle.decrypt_message(k.p, 4, k.s0, ev.BootstrapLUT(cts[1], lut)) == f(1)
    ref = oracle.bootstrap(k.p, k.bsk, k.ksk, cts[2], lut)
    assert oracle.decrypt_message(k.p, 4, k.s0, ref) == f(2)


def test_batch_larger_than_one_launch(oracle, pkg, keys_u2, ck_u2):
    # 8 bootstraps per CU are co-resident; a batch beyond that is issued as chunked launches
    k = keys_u2
    B = 8 * 256 + 37
    rs = np.random.RandomState(46)
    msgs = rs.randint(0, 4, B)
    lut = oracle.lut_generate(k.p, [(x + 1) % 4 for x in range(4)])
    cts = np.stack([oracle.encrypt_message(k.p, k.rng, int(m), 4, k.s0) for m in msgs])
    out = ck_u2.ctx.bootstrap_batch(cts, lut)
    dec = np.array([oracle.decrypt_message(k.p, 4, k.s0, np.ascontiguousarray(o)) for o in out])
    assert np.array_equal(dec, (msgs + 1) % 4)


def test_gates_at_uint2_params(oracle, pkg, keys_u2, ck_u2):
    # the gate prologue of the N=512 kernel (gates.go:26-104 run with CurrentSecurityLevel = Uint2)
    k = keys_u2
    bits = [(0, 0), (0, 1), (1, 0), (1, 1)]
    a = np.stack([oracle.encrypt_bool(k.p, k.rng, x, k.s0) for x, _ in bits])
    b = np.stack([oracle.encrypt_bool(k.p, k.rng, y, k.s0) for _, y in bits])
    for op, f in (("NAND", lambda x, y: 1 - (x & y)), ("XOR", lambda x, y: x ^ y), ("OR", lambda x, y: x | y)):
        out = ck_u2.ctx.gate_batch(op, a, b)
        assert [oracle.decrypt_bool(k.p, k.s0, np.ascontiguousarray(o)) for o in out] == [f(x, y) for x, y in bits], op
