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


def test_default_seed_is_os_entropy(oracle, pkg):
    # CloudKey.NewCloudKey(seed=None) -> tfhe_keygen_cloud_seeded(..., NULL): 128 bits from getrandom per call, like the
    # reference's auto-seeded generator (key/key.go:17).  Two keys from the same secret key differ (their blobs share
    # essentially no words), both evaluate gates correctly; a fixed 128-bit seed is reproducible and differs from the
    # 64-bit seed with the same low word.
    p = oracle.params("128").small(12)
    rng = oracle.rng(0x7F4E0045)
    s0, s1 = oracle.keygen_secret(p, rng)
    A, B = [0, 0, 1, 1], [0, 1, 0, 1]
    a, b = oracle.encrypt_bools(p, rng, A, s0), oracle.encrypt_bools(p, rng, B, s0)
    blobs = []
    for seed in (None, None, (5 << 64) | 9, (5 << 64) | 9, 9):
        ck = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=seed)
        assert list(oracle.decrypt_bools(p, s0, ck.ctx.gate_batch("NAND", a, b))) == [True, True, True, False]
        blobs.append(ck.ctx.key_export_dev(0).cpu().numpy()[64:].view(np.uint32).copy())   # the bootstrapping key behind the 64-byte blob header (no padding words)
        ck.close()
    assert (blobs[0] == blobs[1]).mean() < 0.01                   # OS entropy: unrelated keys
    assert np.array_equal(blobs[2], blobs[3])                     # fixed seed: reproducible
    assert (blobs[2] == blobs[4]).mean() < 0.01                   # the high 64 bits of the seed matter
    with pytest.raises(ValueError):
        pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=1 << 128)


def test_key_blobs_through_host_memory(oracle, pkg, tmp_path):
    # tfhe_key_export / tfhe_key_import: a GPU-generated cloud key saved to disk and loaded into a fresh context (any
    # N: here the N = 512 ring of Uint2 and the 128-bit gate set) evaluates identically, word for word.
    for name in ("128", "uint2"):
        p = oracle.params(name).small(10)
        rng = oracle.rng(0x7F4E0046)
        s0, s1 = oracle.keygen_secret(p, rng)
        ck = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=77)
        cts = np.stack([oracle.encrypt_message(p, rng, m % 2, 2, s0) for m in range(6)])
        lut = oracle.lut_generate(p, [1, 0])
        want = ck.ctx.bootstrap_batch(cts, lut)
        for which in (0, 1):
            np.save(tmp_path / f"{name}_{which}.npy", ck.ctx.key_export(which))
        ck.close()
        ck2 = pkg.CloudKey(gpu_params(pkg, p))
        for which in (0, 1):
            ck2.ctx.key_import(which, np.load(tmp_path / f"{name}_{which}.npy"))
        assert np.array_equal(ck2.ctx.bootstrap_batch(cts, lut), want), name
        # what the header is for: a short buffer, a blob of the other key, a damaged header and a blob of another
        # parameter set are all refused (TFHE_E_INVALID) and install nothing
        good = np.load(tmp_path / f"{name}_0.npy")
        assert good.size == ck2.ctx.key_size(0) and bytes(good[:7]) == b"TFHEKEY"
        bad_param = good.copy(); bad_param[16] ^= 1                      # first parameter word, checksum no longer matches
        for bad in (np.zeros(16, np.uint8), good[:-1], np.load(tmp_path / f"{name}_1.npy"), bad_param, np.zeros_like(good)):
            with pytest.raises(pkg.TfheError):
                ck2.ctx.key_import(0, bad)
        assert np.array_equal(ck2.ctx.bootstrap_batch(cts, lut), want), name     # still the good key
        ck2.close()
    # a blob of one parameter set offered to a context of another (same ring, other LWE dimension)
    p_a, p_b = oracle.params("128").small(10), oracle.params("128").small(12)
    rng = oracle.rng(0x7F4E0047)
    s0, s1 = oracle.keygen_secret(p_a, rng)
    ck_a = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p_a), s0, s1, p_a.alpha_lv0, p_a.alpha_lv1, seed=78)
    ck_b = pkg.CloudKey(gpu_params(pkg, p_b))
    with pytest.raises(pkg.TfheError, match="another parameter set"):
        ck_b.ctx.key_import(0, ck_a.ctx.key_export(0))
    ck_a.close(); ck_b.close()


def _wave_blob_to_reference_spectra(blob, n, L):
    """Device layout cd bsk[n][2][L][2][8][64] (csrc/kernels.hpp) -> reference FourierPoly layout [n][2L][2][1024]:
    (reg, lane) holds root u = (lane>>3) + 8*(lane&7) + 64*reg, the reference keeps it in slot bitrev9(-u mod 512),
    stored as blocks of [4 re | 4 im] (poly.go:57-62)."""
    z = blob.view(np.float64).reshape(n, 2, L, 2, 8, 64, 2)
    reg, lane = np.meshgrid(np.arange(8), np.arange(64), indexing="ij")
    u = (lane >> 3) + 8 * (lane & 7) + 64 * reg
    v = (512 - u) & 511
    slot = np.zeros_like(v)
    for b in range(9):
        slot |= ((v >> b) & 1) << (8 - b)
    base = 8 * (slot >> 2) + (slot & 3)
    out = np.zeros((n, 2, L, 2, 1024), np.float64)
    out[..., base] = z[..., 0]
    out[..., base + 4] = z[..., 1]
    return out.reshape(n, 2 * L, 2, 1024)                  # reference row r = p*L + l


def _centered(x):
    return ((x.astype(np.int64) + 2**31) % 2**32) - 2**31


def test_keygen_noise_statistics(oracle, pkg):
    # The one security-relevant property of tfhe_keygen_cloud that decrypt-level tests cannot see: the noise it adds.
    # The generated keys are read back as device-layout blobs, every ciphertext in them is "decrypted" with the
    # secret keys and the known plaintext removed; what is left must be the prescribed Gaussian (cloudkey.go:88-145:
    # KSK rows at alpha_lv0, TRGSW rows at alpha_lv1), not zero and not something else.
    p = oracle.params("128").small(8)
    rng = oracle.rng(0x7F4E0044)
    s0, s1 = oracle.keygen_secret(p, rng)
    ck = pkg.CloudKey.NewCloudKey(gpu_params(pkg, p), s0, s1, p.alpha_lv0, p.alpha_lv1, seed=2024)
    ksk = ck.ctx.key_export_dev(1).cpu().numpy()[64:].view(np.uint32)          # payloads behind the 64-byte blob headers
    bskb = ck.ctx.key_export_dev(0).cpu().numpy()[64:]
    ck.close()
    # ---- key-switching key: packed rows [N*t*(base-1) + 1][n1p], row (i, j, k-1) encrypts k*s1[i]*2^(32-(j+1)*basebit) under s0
    n1p = (p.n + 1 + 31) & ~31                              # rows are whole 128-byte lines (tfhe_ctx::n1p)
    rows = ksk.reshape(-1, n1p)[:-1]                        # the last row is the all-zero padding row
    base1 = (1 << p.basebit) - 1
    assert rows.shape[0] == p.N * p.t * base1 and rows.shape[0] >= 4096
    r = np.arange(rows.shape[0])
    i, j, k = r // (p.t * base1), (r // base1) % p.t, r % base1 + 1
    msg = (k.astype(np.uint64) * s1[i].astype(np.uint64) << (32 - (j + 1) * p.basebit).astype(np.uint64)) & 0xFFFFFFFF
    phase = (rows[:, p.n].astype(np.uint64) - (rows[:, :p.n].astype(np.uint64) * s0.astype(np.uint64)).sum(1) - msg) & 0xFFFFFFFF
    e0 = _centered(phase).astype(np.float64)
    sigma0 = p.alpha_lv0 * 2.0**32
    assert abs(e0.mean()) < 5 * sigma0 / np.sqrt(e0.size), e0.mean()
    assert 0.85 * sigma0 < e0.std() < 1.15 * sigma0, (e0.std(), sigma0)
    assert 0.64 < np.mean(np.abs(e0) < sigma0) < 0.72            # a Gaussian has 68.3 % within one sigma
    assert not rows[:, n1p - 1].any() or n1p == p.n + 1              # padding words stay zero
    # ---- bootstrapping key: TRGSW row r of key i is (A, B = A*s1 + e) + s0[i]*2^(32-(l+1)*Bgbit) on A (r < L) or B (r >= L)
    spec = _wave_blob_to_reference_spectra(bskb, p.n, p.L)
    polys = np.stack([oracle.to_poly(x) for x in spec.reshape(-1, p.N)]).reshape(p.n, 2 * p.L, 2, p.N)
    s1l = s1.astype(np.int64)
    noise = []
    for ii in range(p.n):
        for rr in range(2 * p.L):
            A, Bp = polys[ii, rr, 0].astype(np.int64), polys[ii, rr, 1].astype(np.int64)
            mu = (int(s0[ii]) << (32 - (rr % p.L + 1) * p.Bgbit)) & 0xFFFFFFFF
            if rr < p.L:
                A[0] = (A[0] - mu) & 0xFFFFFFFF           # the gadget term sits on coefficient 0 of A
            else:
                Bp[0] = (Bp[0] - mu) & 0xFFFFFFFF
            full = np.convolve(A, s1l)                      # exact: |sum| < 2^42
            As = full[:p.N].copy()
            As[:p.N - 1] -= full[p.N:]                      # X^N = -1
            noise.append(_centered((Bp - As) & 0xFFFFFFFF))
    e1 = np.concatenate(noise).astype(np.float64)
    sigma1 = p.alpha_lv1 * 2.0**32                          # 85.9 torus units at the 128-bit set
    assert e1.size >= 4096 and sigma1 > 10
    assert abs(e1.mean()) < 5 * sigma1 / np.sqrt(e1.size) + 0.6, e1.mean()      # + the truncation of F64ToTorus (toward zero)
    assert 0.85 * sigma1 < e1.std() < 1.15 * sigma1, (e1.std(), sigma1)
    assert 0.64 < np.mean(np.abs(e1) < sigma1) < 0.72
