"""GPU tier: the trgsw / trlwe seams of SURVEY.md 8(b) seam 3 behind the C ABI, with CALLER-SUPPLIED operands -- what
shim/go/trgsw, shim/go/trlwe and Evaluator.ExternalProductAssign / CMuxAssign of shim/go/evaluator bind:

  tfhe_external_product_with   trgsw.ExternalProductWithFFT (trgsw/trgsw.go:108-137), Evaluator.ExternalProductAssign (evaluator/evaluator.go:50-81)
  tfhe_cmux_with               trgsw.CMUX (trgsw/trgsw.go:173-194), Evaluator.CMuxAssign (evaluator/evaluator.go:85-106)
  tfhe_sample_extract_batch    trlwe.SampleExtractIndex[Assign] for any k (trlwe/trlwe.go:114-128, trlwe/trlwe_ops.go:10-21)
  tfhe_keyswitch_batch         trgsw.IdentityKeySwitching[Assign] (trgsw/trgsw.go:285-312, trgsw/keyswitch.go:10-37)
  tfhe_ctx_decomposition_offset  cloudkey.go:60-71

Bit-exact against the C oracle at the N = 1024, L = 3, Bgbit = 6 shape (and against the FFT-free exact-integer product); at the Uint5
shape within the stated per-product tolerance (2^9 torus units per coefficient, here against the oracle's fp64 pipeline)."""
import numpy as np
import pytest

from conftest import KeySet, gpu_params, rand_u32

pytestmark = pytest.mark.gpu


def trlwe_batch(rs, B, N):
    return rand_u32(rs, (B, 2, N))


def test_decomposition_offset_is_the_cloud_keys(oracle, keys_small, ck_small, keys80, ck80):
    assert ck_small.ctx.decomposition_offset() == oracle.offset(keys_small.p)
    assert ck80.ctx.decomposition_offset() == oracle.offset(keys80.p)


def test_external_product_with_any_operand_bit_exact(pkg, oracle, keys_small, ck_small):
    k = keys_small
    other = KeySet(oracle, "128", 0x7F4E00A1, n_override=3)            # TRGSW samples that are NOT in the loaded key
    blank = pkg.CloudKey(gpu_params(pkg, k.p))                        # ... and a context that holds no key at all
    try:
        rs = np.random.RandomState(31)
        trl = trlwe_batch(rs, 5, k.p.N)
        for ctx in (ck_small.ctx, blank.ctx):
            for gsw, gsw_t in ((other.bsk[1], other.bsk_torus[1]), (k.bsk[7], k.bsk_torus[7])):
                got = ctx.external_product_with(gsw, trl)
                for b in range(5):
                    assert np.array_equal(got[b], oracle.external_product(k.p, gsw, trl[b]))
                    assert np.array_equal(got[b], oracle.external_product_exact(k.p, gsw_t, trl[b]))
        # an element of the loaded key through both entry points
        assert np.array_equal(ck_small.ctx.external_product_with(k.bsk[7], trl), ck_small.ctx.external_product_batch(7, trl))
        # the explicit default offset is the default
        assert np.array_equal(blank.ctx.external_product_with(other.bsk[0], trl, offset=oracle.offset(k.p)),
                              blank.ctx.external_product_with(other.bsk[0], trl))
    finally:
        blank.close()


def test_external_product_with_a_callers_own_offset(pkg, oracle, keys_small, ck_small):
    # the reference passes decompositionOffset as an argument (trgsw.go:108, evaluator.go:50): any value is a kernel operand here
    k = keys_small
    rs = np.random.RandomState(32)
    trl = trlwe_batch(rs, 3, k.p.N)
    std = oracle.offset(k.p)
    assert np.array_equal(oracle.external_product_at_offset(k.p, k.bsk[2], trl[0], std), oracle.external_product(k.p, k.bsk[2], trl[0]))
    for off in (0, 0x12345678, std ^ 0x80000000):
        got = ck_small.ctx.external_product_with(k.bsk[2], trl, offset=off)
        cm = ck_small.ctx.cmux_with(k.bsk[2], trl, trl[::-1].copy(), offset=off)
        for b in range(3):
            assert np.array_equal(cm[b], oracle.cmux_at_offset(k.p, k.bsk[2], trl[b], trl[2 - b], off)), (off, b)
            assert np.array_equal(got[b], oracle.external_product_at_offset(k.p, k.bsk[2], trl[b], off)), (off, b)


def test_cmux_with_selects_and_is_bit_exact(pkg, oracle, keys_small, ck_small):
    k = keys_small
    rs = np.random.RandomState(33)
    ct0, ct1 = trlwe_batch(rs, 4, k.p.N), trlwe_batch(rs, 4, k.p.N)
    for i in (0, 5, 11):
        got = ck_small.ctx.cmux_with(k.bsk[i], ct0, ct1)
        for b in range(4):
            assert np.array_equal(got[b], oracle.cmux(k.p, k.bsk[i], ct0[b], ct1[b])), (i, b)
    # one CMUX step of a blind rotation IS cmux_with on (acc, X^a acc): prefix of one step through the two entry points
    cts = rand_u32(rs, (2, k.p.n + 1))
    tv = oracle.gate_testvec(k.p)
    acc0 = ck_small.ctx.blind_rotate_batch(cts, nsteps=0)
    acc1 = ck_small.ctx.blind_rotate_batch(cts, nsteps=1)
    for b in range(2):
        a_t = ((int(cts[b, 0]) + (1 << (31 - k.p.Nbit - 1))) & 0xFFFFFFFF) >> (32 - k.p.Nbit - 1)
        rot = np.stack([oracle.poly_mul_xk(np.ascontiguousarray(acc0[b, part]), a_t) for part in range(2)])
        assert np.array_equal(ck_small.ctx.cmux_with(k.bsk[0], acc0[b][None], rot[None])[0], acc1[b])
    assert tv.shape == (2, k.p.N)


def test_sample_extract_any_index(pkg, oracle, keys_small, ck_small):
    k = keys_small
    rs = np.random.RandomState(34)
    trl = trlwe_batch(rs, 3, k.p.N)
    for idx in (0, 1, 511, k.p.N - 1):
        got = ck_small.ctx.sample_extract_batch(trl, idx)
        for b in range(3):
            assert np.array_equal(got[b], oracle.sample_extract(np.ascontiguousarray(trl[b]), idx)), (idx, b)
    for bad in (-1, k.p.N):
        with pytest.raises(pkg.TfheError, match="index"):
            ck_small.ctx.sample_extract_batch(trl, bad)


@pytest.mark.parametrize("B", [1, 7, 40, 300, 1100, 2100])            # 1100, 2100: across the matrix-core key switch's 1,024-ciphertext chunks
def test_keyswitch_on_extracted_samples_equals_the_fused_form_and_the_oracle(oracle, keys_small, ck_small, B):
    k = keys_small
    rs = np.random.RandomState(35 + B)
    trl = trlwe_batch(rs, B, k.p.N)
    lwe1 = ck_small.ctx.sample_extract_batch(trl, 0)
    got = ck_small.ctx.keyswitch_batch(lwe1)
    assert np.array_equal(got, ck_small.ctx.extract_keyswitch_batch(trl))
    for b in range(0, B, max(1, B // 5)):
        assert np.array_equal(got[b], oracle.key_switch(k.p, k.ksk, lwe1[b]))
    # an arbitrary TLWELv1 (not an extraction of anything the engine made)
    free = rand_u32(rs, (min(B, 9), k.p.N + 1))
    got = ck_small.ctx.keyswitch_batch(free)
    for b in range(free.shape[0]):
        assert np.array_equal(got[b], oracle.key_switch(k.p, k.ksk, free[b]))


def test_seams_at_the_uint5_shape_within_tolerance(pkg, oracle):
    # N = 2048, L = 1, Bgbit = 22: transforms are not exact; one external product within 2^9 of the oracle's
    k = KeySet(oracle, "uint5", 0x7F4E00A5, n_override=4, torus=False)
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    try:
        rs = np.random.RandomState(36)
        trl = trlwe_batch(rs, 3, k.p.N)
        got = ck.ctx.external_product_with(k.bsk[1], trl)
        assert np.array_equal(got, ck.ctx.external_product_batch(1, trl))                       # same kernel, same operand: same words
        for b in range(3):
            want = oracle.external_product(k.p, k.bsk[1], trl[b])
            d = (got[b].astype(np.int64) - want.astype(np.int64) + 2**31) % 2**32 - 2**31
            assert np.abs(d).max() <= 2**9
        cm = ck.ctx.cmux_with(k.bsk[2], trl, trl[::-1].copy())
        for b in range(3):
            want = oracle.cmux(k.p, k.bsk[2], trl[b], trl[2 - b])
            d = (cm[b].astype(np.int64) - want.astype(np.int64) + 2**31) % 2**32 - 2**31
            assert np.abs(d).max() <= 2**9
        lwe1 = ck.ctx.sample_extract_batch(trl, 0)
        assert np.array_equal(lwe1[1], oracle.sample_extract(np.ascontiguousarray(trl[1]), 0))
        assert np.array_equal(ck.ctx.keyswitch_batch(lwe1), ck.ctx.extract_keyswitch_batch(trl))   # integer path: exact at every shape
        assert np.array_equal(ck.ctx.keyswitch_batch(lwe1)[2], oracle.key_switch(k.p, k.ksk, lwe1[2]))
    finally:
        ck.close()


def test_seam_argument_errors(pkg, keys_small, ck_small):
    k = keys_small
    trl = np.zeros((1, 2, k.p.N), np.uint32)
    with pytest.raises(ValueError):
        ck_small.ctx.external_product_with(np.zeros(7), trl)
    nokey = pkg.CloudKey(gpu_params(pkg, k.p))
    try:
        with pytest.raises(pkg.TfheError, match="not loaded"):
            nokey.ctx.keyswitch_batch(np.zeros((1, k.p.N + 1), np.uint32))
        assert nokey.ctx.sample_extract_batch(trl, 3).shape == (1, k.p.N + 1)               # needs no key
        assert nokey.ctx.keyswitch_batch(np.zeros((0, k.p.N + 1), np.uint32)).shape == (0, k.p.n + 1)
    finally:
        nokey.close()


def test_python_mirror_of_the_trgsw_and_trlwe_packages(pkg, oracle, keys_small, ck_small):
    # go-tfhe_amd/trgsw.py, trlwe.py and Evaluator.ExternalProductAssign / CMuxAssign: the reference's names and argument order (host mirror)
    k = keys_small
    rs = np.random.RandomState(37)
    t0, t1 = trlwe_batch(rs, 1, k.p.N)[0], trlwe_batch(rs, 1, k.p.N)[0]
    off = oracle.offset(k.p)
    assert np.array_equal(pkg.trgsw.ExternalProductWithFFT(k.bsk[3], t0, off, ck_small), oracle.external_product(k.p, k.bsk[3], t0))
    assert np.array_equal(pkg.trgsw.CMUX(t0, t1, k.bsk[3], off, ck_small), oracle.cmux(k.p, k.bsk[3], t0, t1))
    ev = pkg.evaluator.Evaluator(ck_small)
    out = np.empty_like(t0)
    ev.ExternalProductAssign(k.bsk[3], t0, out, off)
    assert np.array_equal(out, oracle.external_product(k.p, k.bsk[3], t0))
    ev.ExternalProductAssign(3, t0, out)                                          # by index into the resident key
    assert np.array_equal(out, oracle.external_product(k.p, k.bsk[3], t0))
    ev.CMuxAssign(k.bsk[3], t0, t1, out, off)
    assert np.array_equal(out, oracle.cmux(k.p, k.bsk[3], t0, t1))
    cts = k.enc([1, 0, 1])
    accs = pkg.trgsw.BatchBlindRotate(cts, k.tv, off, ck_small)
    assert np.array_equal(accs[1], oracle.blind_rotate(k.p, k.bsk, cts[1], k.tv))
    assert np.array_equal(pkg.trgsw.BlindRotate(cts[2], k.tv, off, ck_small), accs[2])
    with pytest.raises(ValueError, match="decompositionOffset"):
        pkg.trgsw.BlindRotate(cts[0], k.tv, off + 1, ck_small)
    ext = pkg.trlwe.SampleExtractIndex(accs[0], 0, ck_small)
    assert np.array_equal(ext, oracle.sample_extract(np.ascontiguousarray(accs[0]), 0))
    lv0 = pkg.trgsw.IdentityKeySwitching(ext, ck_small)
    assert np.array_equal(lv0, oracle.bootstrap(k.p, k.bsk, k.ksk, cts[0], k.tv))
