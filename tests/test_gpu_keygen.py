"""GPU tier: cloud-key generation on the GPU (tfhe_keygen_cloud; cloudkey.NewCloudKey, cloudkey.go:24-31).
Key generation is randomised, so parity is at the decrypt level (as in the reference's own tests):
gates evaluated under a GPU-generated cloud key decrypt correctly with the CPU oracle's
encrypt/decrypt under the same secret key, and a noise-free key reproduces the ideal phases."""
import numpy as np
import pytest

from conftest import gpu_params

pytestmark = pytest.mark.gpu


def circ_dist(a, b):
    d = (np.asarray(a, np.int64) - np.asarray(b, np.int64)) % 2**32
    return np.minimum(d, 2**32 - d)


@pytest.mark.parametrize("name", ["80", "128"])
def test_gpu_generated_key_gates(oracle, pkg, name):
    p = oracle.params(name)
    rng = oracle.rng(0x7F4E0031)
    s0, s1 = oracle.keygen_secret(p, rng)
    ck = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=1234)
    A, B = [0, 0, 1, 1] * 16, [0, 1, 0, 1] * 16
    a, b = oracle.encrypt_bools(p, rng, A, s0), oracle.encrypt_bools(p, rng, B, s0)
    for op, f in (("NAND", lambda x, y: not (x and y)), ("XOR", lambda x, y: x != y), ("ORNY", lambda x, y: (not x) or y)):
        out = ck.ctx.gate_batch(op, a, b)
        assert list(oracle.decrypt_bools(p, s0, out)) == [bool(f(bool(x), bool(y))) for x, y in zip(A, B)], op
        ph = np.array([oracle.phase(p, s0, np.ascontiguousarray(o)) for o in out])
        ideal = np.where(oracle.decrypt_bools(p, s0, out), 0x20000000, 0xE0000000)
        assert circ_dist(ph, ideal).max() < 2**27, op          # well inside the +-1/8 decision margin
    c = oracle.encrypt_bools(p, rng, [1, 0] * 32, s0)
    out = ck.ctx.gate_batch("MUX", a, b, c)
    assert list(oracle.decrypt_bools(p, s0, out)) == [bool(y if x else z) for x, y, z in zip(A, B, [1, 0] * 32)]
    # same (seed, key) -> same cloud key -> identical ciphertexts; another seed -> different masks, same bits
    ck2 = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=1234)
    assert np.array_equal(ck2.ctx.gate_batch("NAND", a, b), ck.ctx.gate_batch("NAND", a, b))
    ck3 = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=99)
    o3 = ck3.ctx.gate_batch("NAND", a, b)
    assert not np.array_equal(o3, ck.ctx.gate_batch("NAND", a, b))
    assert list(oracle.decrypt_bools(p, s0, o3)) == [not (x and y) for x, y in zip(A, B)]
    for k in (ck, ck2, ck3):
        k.close()


def test_noise_free_key_matches_oracle_noise_floor(oracle, pkg):
    # alpha = 0: what is left is the gadget / key-switch TRUNCATION error of the reference algorithm
    # (decomposer.go:55-66 has no rounding term), the same for a CPU-generated and a GPU-generated key.
    p = oracle.params("128")
    rng = oracle.rng(0x7F4E0032)
    s0, s1 = oracle.keygen_secret(p, rng)
    A, B = [1, 0, 1, 1] * 8, [1, 1, 0, 1] * 8
    a, b = oracle.encrypt_bools(p, rng, A, s0), oracle.encrypt_bools(p, rng, B, s0)
    ideal = np.where(np.array(A) & np.array(B), 0x20000000, 0xE0000000)
    ck = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, 0.0, 0.0, seed=5)
    out = ck.ctx.gate_batch("AND", a, b)
    dev_gpu = circ_dist([oracle.phase(p, s0, np.ascontiguousarray(o)) for o in out], ideal)
    ck.close()
    # the oracle harness with the same secret key and alpha = 0
    q = oracle.params("128")
    q.alpha_lv0 = 0.0
    q.alpha_lv1 = 0.0
    _, bsk = oracle.keygen_bsk(q, rng, s0, s1, torus=False)
    ksk = oracle.keygen_ksk(q, rng, s0, s1)
    ck2 = pkg.CloudKey(gpu_params(pkg, p), bsk_fourier=bsk, ksk=ksk)
    out2 = ck2.ctx.gate_batch("AND", a, b)
    dev_cpu = circ_dist([oracle.phase(p, s0, np.ascontiguousarray(o)) for o in out2], ideal)
    ck2.close()
    assert dev_gpu.max() < 2**27 and dev_cpu.max() < 2**27
    assert 1.0 / 3.0 < dev_gpu.mean() / dev_cpu.mean() < 3.0, (dev_gpu.mean(), dev_cpu.mean())


def test_gpu_generated_key_uint5_pbs(oracle, pkg):
    p = oracle.params("uint5").small(64)
    rng = oracle.rng(0x7F4E0033)
    s0, s1 = oracle.keygen_secret(p, rng)
    ck = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=7)
    lut = oracle.lut_generate(p, [(3 * x + 1) % 32 for x in range(32)])
    msgs = [0, 1, 5, 15, 16, 30, 31]
    cts = np.stack([oracle.encrypt_message(p, rng, m, 32, s0) for m in msgs])
    out = ck.ctx.bootstrap_batch(cts, lut)
    assert [oracle.decrypt_message(p, 32, s0, np.ascontiguousarray(o)) for o in out] == [(3 * m + 1) % 32 for m in msgs]
    ck.close()


def test_keygen_rejects_bad_input(pkg):
    p = pkg.params.Security80Bit
    ctx = pkg.Context(p)
    s0, s1 = np.zeros(p.n, np.uint32), np.zeros(p.N, np.uint32)
    s0[3] = 2
    with pytest.raises(pkg.TfheError):
        ctx.keygen_cloud(s0, s1, 1e-5, 1e-8, 1)
    ctx.close()
