"""GPU parity for the hot path: external product, CMUX chain, key switch, bootstrap, gates.
Integer torus outputs are compared BIT-EXACT with the oracle (and the exact-integer oracle)."""
import numpy as np
import pytest

from conftest import gpu_params, rand_u32

pytestmark = pytest.mark.gpu

ALL_OPS = ["NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN"]
TRUTH = {
    "NAND": lambda a, b: not (a and b), "AND": lambda a, b: a and b, "OR": lambda a, b: a or b,
    "XOR": lambda a, b: a != b, "XNOR": lambda a, b: a == b, "NOR": lambda a, b: not (a or b),
    "ANDNY": lambda a, b: (not a) and b, "ANDYN": lambda a, b: a and (not b),
    "ORNY": lambda a, b: (not a) or b, "ORYN": lambda a, b: a or (not b),
}


def test_external_product_bit_exact(oracle, keys_small, ck_small):
    k = keys_small
    rs = np.random.RandomState(10)
    trl = rand_u32(rs, (6, 2, 1024))
    trl[0] = 0                                  # zero input -> zero digits -> zero output
    for idx in (0, 7, k.p.n - 1):
        got = ck_small.ctx.external_product_batch(idx, trl)
        for b in range(trl.shape[0]):
            assert np.array_equal(got[b], oracle.external_product(k.p, k.bsk[idx], trl[b])), (idx, b)
            assert np.array_equal(got[b], oracle.external_product_exact(k.p, k.bsk_torus[idx], trl[b])), (idx, b)
    assert not ck_small.ctx.external_product_batch(0, trl)[0].any()


def test_bsk_torus_upload_equals_fourier_upload(pkg, oracle, keys_small):
    from conftest import gpu_params
    k = keys_small
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_torus=k.bsk_torus, ksk=k.ksk)
    rs = np.random.RandomState(11)
    trl = rand_u32(rs, (3, 2, 1024))
    got = ck.ctx.external_product_batch(5, trl)
    for b in range(3):
        assert np.array_equal(got[b], oracle.external_product_exact(k.p, k.bsk_torus[5], trl[b]))
    ck.close()


@pytest.mark.parametrize("nsteps", [0, 1, 2, 9, -1])
def test_blind_rotate_prefix_bit_exact(oracle, keys_small, ck_small, nsteps):
    k = keys_small
    rs = np.random.RandomState(12)
    cts = rand_u32(rs, (5, k.p.n + 1))
    cts[0] = 0                                   # b = 0 -> btilde = 2N (wraps to 0), all atilde = 0
    cts[1] = 0xFFFFFFFF                          # extreme mod-switch inputs
    got = ck_small.ctx.blind_rotate_batch(cts, None, nsteps)
    for b in range(cts.shape[0]):
        want = oracle.blind_rotate(k.p, k.bsk, cts[b], k.tv, nsteps)
        assert np.array_equal(got[b], want), (nsteps, b)


def test_blind_rotate_per_item_testvec(oracle, keys_small, ck_small):
    k = keys_small
    rs = np.random.RandomState(13)
    cts = rand_u32(rs, (3, k.p.n + 1))
    tvs = rand_u32(rs, (3, 2, 1024))
    got = ck_small.ctx.blind_rotate_batch(cts, tvs)
    for b in range(3):
        assert np.array_equal(got[b], oracle.blind_rotate(k.p, k.bsk, cts[b], tvs[b]))
    got1 = ck_small.ctx.blind_rotate_batch(cts, tvs[0])
    for b in range(3):
        assert np.array_equal(got1[b], oracle.blind_rotate(k.p, k.bsk, cts[b], tvs[0]))


def test_blind_rotate_vs_exact_integer_oracle(oracle, keys_small, ck_small):
    k = keys_small
    rs = np.random.RandomState(14)
    cts = rand_u32(rs, (2, k.p.n + 1))
    got = ck_small.ctx.blind_rotate_batch(cts)
    for b in range(2):
        assert np.array_equal(got[b], oracle.blind_rotate_exact(k.p, k.bsk_torus, cts[b], k.tv))


@pytest.mark.parametrize("which", ["small", "80", "128"])
def test_extract_keyswitch_bit_exact(oracle, request, which):
    k = request.getfixturevalue({"small": "keys_small", "80": "keys80", "128": "keys128"}[which])
    ck = request.getfixturevalue({"small": "ck_small", "80": "ck80", "128": "ck128"}[which])
    rs = np.random.RandomState(15)
    trl = rand_u32(rs, (5, 2, 1024))
    trl[0] = 0                                   # all digits: a_i = ~0 except a_0 = 0
    trl[1] = 0xFFFFFFFF
    got = ck.ctx.extract_keyswitch_batch(trl)
    for b in range(trl.shape[0]):
        want = oracle.key_switch(k.p, k.ksk, oracle.sample_extract(trl[b]))
        assert np.array_equal(got[b], want), b


@pytest.mark.parametrize("which,B", [("small", 70), ("128", 33), ("80", 64)])
def test_extract_keyswitch_tiled_kernel_bit_exact(oracle, request, which, B):
    # B >= 32 takes the tiled base-4 kernel (ragged last tile when B % 32 != 0)
    k = request.getfixturevalue({"small": "keys_small", "80": "keys80", "128": "keys128"}[which])
    ck = request.getfixturevalue({"small": "ck_small", "80": "ck80", "128": "ck128"}[which])
    rs = np.random.RandomState(19)
    trl = rand_u32(rs, (B, 2, 1024))
    trl[3] = 0
    trl[4] = 0xFFFFFFFF
    got = ck.ctx.extract_keyswitch_batch(trl)
    for b in range(B):
        want = oracle.key_switch(k.p, k.ksk, oracle.sample_extract(trl[b]))
        assert np.array_equal(got[b], want), b


@pytest.mark.parametrize("which", ["small", "80", "128", "uint2"])
def test_matrix_core_keyswitch_equals_vector_kernels(pkg, oracle, request, which):
    # csrc/keyswitch_mfma.hpp (the default for the base-4 sets: exact int8 matrix product over byte columns) against the
    # vector-ALU kernels of csrc/kernels.hpp (option ks_mfma_min = 0: per-ciphertext gather below 32, the tiled kernel above)
    # on the same key and inputs, bit for bit, across the tile edges (256-row groups, 1,024-row chunks) -- and both
    # against the oracle on a sample.
    # "uint2": base 16 (one coefficient's 16 candidate rows per 16-K piece), N = 512, a random key of reduced dimension.
    if which == "uint2":
        from types import SimpleNamespace
        p2 = oracle.params("uint2").small(40)
        k = SimpleNamespace(p=p2, ksk=rand_u32(np.random.RandomState(29), (p2.N * p2.t * (1 << p2.basebit), p2.n + 1)))
    else:
        k = request.getfixturevalue({"small": "keys_small", "80": "keys80", "128": "keys128"}[which])
    ckv = pkg.CloudKey(gpu_params(pkg, k.p), ksk=k.ksk)
    ckv.ctx.set_option("ks_mfma_min", 0)
    ckm = pkg.CloudKey(gpu_params(pkg, k.p), ksk=k.ksk)
    assert ckm.ctx.get_option("ks_mfma_min") == 24 and ckv.ctx.get_option("ks_mfma_min") == 0      # default: matrix cores from 24 ciphertexts on
    ckm.ctx.set_option("ks_mfma_min", 1)                                                            # here: always
    rs = np.random.RandomState(23)
    for B in ((1, 5, 33, 256, 257, 1025, 2100) if which in ("small", "uint2") else (1, 33, 300)):
        trl = rand_u32(rs, (B, 2, k.p.N))
        trl[0] = 0
        trl[B // 2] = 0xFFFFFFFF
        a, b = ckv.ctx.extract_keyswitch_batch(trl), ckm.ctx.extract_keyswitch_batch(trl)
        assert np.array_equal(a, b), (which, B)
        for i in sorted({0, B // 2, B - 1}):
            assert np.array_equal(b[i], oracle.key_switch(k.p, k.ksk, oracle.sample_extract(trl[i]))), (which, B, i)
    ckv.close(); ckm.close()


def test_bootstrap_80bit_bit_exact_and_decrypts(oracle, keys80, ck80):
    k = keys80
    bits = [0, 1, 1]
    cts = k.enc(bits)
    got = ck80.ctx.bootstrap_batch(cts)
    for b in range(len(bits)):
        assert np.array_equal(got[b], oracle.bootstrap(k.p, k.bsk, k.ksk, cts[b], k.tv))
    assert list(k.dec(got)) == [bool(x) for x in bits]


@pytest.mark.parametrize("op", ALL_OPS)
def test_gate_truth_tables_128bit(oracle, keys128, ck128, op):
    # mirrors gates/gates_test.go:23-366 (default = 128-bit parameters)
    k = keys128
    A, B = [0, 0, 1, 1], [0, 1, 0, 1]
    a, b = k.enc(A), k.enc(B)
    out = ck128.ctx.gate_batch(op, a, b)
    assert list(k.dec(out)) == [bool(TRUTH[op](bool(x), bool(y))) for x, y in zip(A, B)]
    # bit-exact vs the oracle's prepare + bootstrap for one of the four
    want = oracle.gate(k.p, k.bsk, k.ksk, op, a[2], b[2])
    assert np.array_equal(out[2], want)


def test_mux_truth_table_128bit(oracle, keys128, ck128, pkg):
    k = keys128
    A = [0, 0, 0, 0, 1, 1, 1, 1]; B = [0, 0, 1, 1, 0, 0, 1, 1]; C = [0, 1, 0, 1, 0, 1, 0, 1]
    a, b, c = k.enc(A), k.enc(B), k.enc(C)
    out = ck128.ctx.gate_batch("MUX", a, b, c)
    assert list(k.dec(out)) == [bool(y if x else z) for x, y, z in zip(A, B, C)]
    assert np.array_equal(out[5], oracle.gate(k.p, k.bsk, k.ksk, "MUX", a[5], b[5], c[5]))
    # scalar API (gates.MUX, gates.go:107-114)
    assert np.array_equal(pkg.gates.MUX(a[5], b[5], c[5], ck128), out[5])


def test_mixed_gate_stream_small(oracle, keys_small, ck_small, pkg):
    # per-item op codes incl. MUX; bit-exact against the oracle on random (noise-free-ish) samples
    k = keys_small
    rs = np.random.RandomState(16)
    names = ["AND", "OR", "XOR", "MUX", "NAND", "MUX", "ORYN", "XNOR", "MUX"]
    B = len(names)
    a, b, c = (rand_u32(rs, (B, k.p.n + 1)) for _ in range(3))
    got = pkg.gates.gate_stream(names, a, b, ck_small, c)
    want, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, [pkg.OPS[x] for x in names], a, b, c)
    assert np.array_equal(got, want)


def test_batch_gates_api(oracle, keys_small, ck_small, pkg):
    # gates.Batch* (gates.go:156-312); BatchXNOR follows scalar XNOR
    k = keys_small
    rs = np.random.RandomState(17)
    pairs = [(rand_u32(rs, k.p.n + 1), rand_u32(rs, k.p.n + 1)) for _ in range(7)]
    for name in ["NAND", "AND", "OR", "XOR", "NOR", "XNOR"]:
        got = getattr(pkg.gates, "Batch" + name)(pairs, ck_small)
        for (x, y), g in zip(pairs, got):
            assert np.array_equal(g, oracle.gate(k.p, k.bsk, k.ksk, name, x, y)), name
    one = pkg.gates.NAND(pairs[0][0], pairs[0][1], ck_small)
    assert np.array_equal(one, oracle.gate(k.p, k.bsk, k.ksk, "NAND", *pairs[0]))


def test_full_size_batch_1024_nand_128bit(oracle, keys128, ck128):
    # BASELINE config 2 at full size: all 1024 decrypt correctly; a sample is bit-exact
    k = keys128
    rs = np.random.RandomState(18)
    A = rs.randint(0, 2, 1024); B = rs.randint(0, 2, 1024)
    a, b = k.enc(A), k.enc(B)
    out = ck128.ctx.gate_batch("NAND", a, b)
    assert np.array_equal(k.dec(out), ~(A.astype(bool) & B.astype(bool)))
    idx = [0, 511, 1023]
    want, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, "NAND", a[idx], b[idx])
    assert np.array_equal(out[idx], want)
    # size-independent property: outputs are deterministic (idempotent launch)
    assert np.array_equal(ck128.ctx.gate_batch("NAND", a, b), out)


@pytest.mark.parametrize("B", [700, 1024, 5000])
def test_batch_position_does_not_matter_128bit(keys128, ck128, B):
    # size-independent property at and above BASELINE config 2's size: a bootstrap's output words depend on its own inputs
    # only -- permuting the batch permutes the outputs bit for bit, and a slice of the batch run on its own gives the same
    # words.  Items change workgroup, partner item, launch (5000 = four full launches + a 904-item one) and, between the
    # full batch and the slice, kernel shape (two free-running workgroups per CU / three / the eight-wave kernel).
    k = keys128
    rs = np.random.RandomState(4100 + B)
    pool_a, pool_b = k.enc(rs.randint(0, 2, 64)), k.enc(rs.randint(0, 2, 64))
    ia, ib = rs.randint(0, 64, B), rs.randint(0, 64, B)
    a, b = pool_a[ia], pool_b[ib]
    out = ck128.ctx.gate_batch("XOR", a, b)
    perm = rs.permutation(B)
    assert np.array_equal(ck128.ctx.gate_batch("XOR", a[perm], b[perm]), out[perm])
    cut = slice(B // 3, B // 3 + 200)
    assert np.array_equal(ck128.ctx.gate_batch("XOR", a[cut], b[cut]), out[cut])
    # equal inputs give equal outputs wherever they sit in the batch
    first = {}
    for i, key in enumerate(zip(ia, ib)):
        j = first.setdefault(key, i)
        if j != i:
            assert np.array_equal(out[i], out[j]), (i, j)


def test_edge_cases_and_errors(pkg, keys_small, ck_small):
    from conftest import gpu_params
    k = keys_small
    n1 = k.p.n + 1
    empty = np.empty((0, n1), np.uint32)
    assert ck_small.ctx.gate_batch("NAND", empty, empty).shape == (0, n1)
    assert ck_small.ctx.bootstrap_batch(empty).shape == (0, n1)
    with pytest.raises(pkg.TfheError):                      # unsupported shape
        pkg.Context(pkg.Params(n=10, N=512, Nbit=9, L=3, Bgbit=6, basebit=2, t=7))
    bare = pkg.Context(gpu_params(pkg, k.p))
    with pytest.raises(pkg.TfheError) as e:                 # no key loaded
        bare.bootstrap_batch(np.zeros((1, n1), np.uint32))
    assert e.value.code == -2
    bare.close()
    with pytest.raises(pkg.TfheError):                      # MUX without third operand
        ck_small.ctx.gate_batch("MUX", np.zeros((1, n1), np.uint32), np.zeros((1, n1), np.uint32))
    with pytest.raises(pkg.TfheError):                      # bad op code
        ck_small.ctx.gate_batch(np.array([77], np.uint8), np.zeros((1, n1), np.uint32), np.zeros((1, n1), np.uint32))


def test_110bit_parameter_set(oracle, pkg):
    # params.go:117-146 (n=630, t=8): same kernels, different LWE dimension / key-switch depth
    from conftest import KeySet, gpu_params
    k = KeySet(oracle, "110", 0x7F4E0006, torus=False)
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    A, B = [0, 0, 1, 1] * 9, [0, 1, 0, 1] * 9                 # 36 items: tiled key switch + ragged tile
    a, b = k.enc(A), k.enc(B)
    for op in ("NAND", "XOR"):
        out = ck.ctx.gate_batch(op, a, b)
        assert list(k.dec(out)) == [bool(TRUTH[op](bool(x), bool(y))) for x, y in zip(A, B)]
        assert np.array_equal(out[33], oracle.gate(k.p, k.bsk, k.ksk, op, a[33], b[33]))
    ck.close()


def test_external_product_adversarial_extreme_within_one_ulp(pkg, oracle):
    # with every digit = -32 and every key coefficient = -2^31 the fp64 pipeline's
    # pre-rounding error reaches ~0.5, so results may be off by one torus ulp from the exact integer
    # product -- for the reference's own FFT as well as for this one.  Random inputs stay bit-exact.
    from conftest import gpu_params
    p = oracle.params("128").small(1)
    bsk_t = np.full((1, 6, 2, 1024), 0x80000000, np.uint32)
    ksk = np.zeros((p.ksk_rows, p.n + 1), np.uint32)
    ck = pkg.CloudKey(gpu_params(pkg, p), bsk_torus=bsk_t, ksk=ksk)
    off = oracle.offset(p)
    trl = np.full((1, 2, 1024), (0 - off) & 0xFFFFFFFF, np.uint32)       # d + offset = 0 -> every digit is -32
    assert (oracle.decompose(p, trl[0][0]).view(np.int32) == -32).all()
    got = ck.ctx.external_product_batch(0, trl)[0]
    exact = oracle.external_product_exact(p, bsk_t[0], trl[0])
    d = (got.astype(np.int64) - exact.astype(np.int64)) % 2**32
    assert np.minimum(d, 2**32 - d).max() <= 1
    bf = np.stack([oracle.to_fourier(x) for x in bsk_t[0].reshape(-1, 1024)]).reshape(6, 2, 1024)
    ref = oracle.external_product(p, bf, trl[0])
    d = (ref.astype(np.int64) - exact.astype(np.int64)) % 2**32
    assert np.minimum(d, 2**32 - d).max() <= 1
    ck.close()


def test_pinned_host_buffers(oracle, keys_small, ck_small, pkg):
    # tfhe_host_alloc / tfhe_host_free: page-locked operands and outputs through the host-pointer ABI
    k = keys_small
    rs = np.random.RandomState(23)
    B = 40
    a, b = rand_u32(rs, (B, k.p.n + 1)), rand_u32(rs, (B, k.p.n + 1))
    pa, pb, po = pkg.PinnedArray(a.shape), pkg.PinnedArray(a.shape), pkg.PinnedArray(a.shape)
    pa.array[...] = a
    pb.array[...] = b
    ck_small.ctx.gate_batch("XOR", pa.array, pb.array, out=po.array)
    want, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, "XOR", a, b)
    assert np.array_equal(po.array, want)
    for x in (pa, pb, po):
        x.free()


def test_batch_larger_than_one_launch_with_per_item_testvec(oracle, keys_small, ck_small):
    # > 1024 items are issued as several launches (launch_blind_rotate): operand / op / test-vector offsets
    k = keys_small
    rs = np.random.RandomState(24)
    B = 1024 + 77
    cts = rand_u32(rs, (B, k.p.n + 1))
    tvs = rand_u32(rs, (B, 2, 1024))
    got = ck_small.ctx.blind_rotate_batch(cts, tvs)
    for b in (0, 1023, 1024, 1025, B - 1):
        assert np.array_equal(got[b], oracle.blind_rotate(k.p, k.bsk, cts[b], tvs[b])), b
    out = ck_small.ctx.bootstrap_batch(cts, tvs)
    for b in (5, 1024, B - 1):
        assert np.array_equal(out[b], oracle.bootstrap(k.p, k.bsk, k.ksk, cts[b], tvs[b])), b


@pytest.mark.parametrize("B", [257, 301, 512])
def test_paired_workgroup_launch_bit_exact(oracle, keys_small, ck_small, pkg, B):
    # 256 < B <= 512 is launched as 4-wave workgroups holding two bootstraps each (k_blind_rotate<.., 2>);
    # an odd B leaves one wave pair idle in the last workgroup.  Per-item test vectors, per-item gate ops.
    k = keys_small
    rs = np.random.RandomState(25 + B)
    cts = rand_u32(rs, (B, k.p.n + 1))
    tvs = rand_u32(rs, (B, 2, 1024))
    got = ck_small.ctx.blind_rotate_batch(cts, tvs)
    for b in (0, 1, 2, 255, 256, B - 2, B - 1):
        assert np.array_equal(got[b], oracle.blind_rotate(k.p, k.bsk, cts[b], tvs[b])), b
    a, c = rand_u32(rs, (B, k.p.n + 1)), rand_u32(rs, (B, k.p.n + 1))
    names = np.array(["NAND", "XOR", "ORNY", "AND"])[rs.randint(0, 4, B)]
    ops = np.array([pkg.OPS[x] for x in names], np.uint8)
    out = ck_small.ctx.gate_batch(ops, a, c)
    sample = [0, 1, 256, B - 1]
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops[sample], np.ascontiguousarray(a[sample]), np.ascontiguousarray(c[sample]))
    assert np.array_equal(out[sample], ref)


@pytest.mark.parametrize("B", [1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 513, 767, 768, 769, 770, 1021, 1023, 1025, 2049])
def test_dispatch_boundaries_gate_shape(oracle, keys_small, ck_small, pkg, B):
    # Every batch-size threshold of the launchers, both sides: gather / tiled key switch (32), the four-wave
    # kernel up to one workgroup per CU (256), two items per workgroup (257..512), one (513..768), four (769..1024, ragged last workgroup
    # with 1..3 idle wave pairs), chunked launches beyond 1024.
    # Whole gates (prep + blind rotate + extract + key switch) are bit-exact against the oracle.
    k = keys_small
    rs = np.random.RandomState(1000 + B)
    a, b = rand_u32(rs, (B, k.p.n + 1)), rand_u32(rs, (B, k.p.n + 1))
    names = np.array(["NAND", "OR", "XNOR", "ANDYN"])[rs.randint(0, 4, B)]
    ops = np.array([pkg.OPS[x] for x in names], np.uint8)
    out = ck_small.ctx.gate_batch(ops, a, b)
    sample = sorted(set([0, B // 2, max(0, B - 2), B - 1]))
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops[sample], np.ascontiguousarray(a[sample]), np.ascontiguousarray(b[sample]))
    assert np.array_equal(out[sample], ref), B
    # a uniform-op launch of the same inputs agrees with the per-item-op launch where the ops coincide
    uni = ck_small.ctx.gate_batch("NAND", a, b)
    sel = np.where(names == "NAND")[0]
    assert np.array_equal(uni[sel], out[sel])


@pytest.mark.parametrize("which", ["small", "uint1", "uint3"])
def test_four_wave_kernel_equals_two_wave_kernel(pkg, oracle, keys_small, which):
    # kernels_quad.hpp (four waves per bootstrap; eight for L = 3 at <= one bootstrap per CU) against kernels.hpp on the same key and inputs: the
    # accumulators are bit-identical word for word at every prefix of the CMUX chain, for all three N = 1024
    # gadget shapes (L=3/Bg=2^6: exact regime, whole chains; Uint1 L=2/Bg=2^10 and Uint3 L=1/Bg=2^23: tolerance
    # regime, one product).
    if which == "small":
        p, bsk_t = keys_small.p, keys_small.bsk_torus
    else:
        p = oracle.params(which).small(16)
        bsk_t = rand_u32(np.random.RandomState(5), (p.n, 2 * p.L, 2, p.N))
    ksk = np.zeros((p.N * p.t * (1 << p.basebit), p.n + 1), np.uint32)
    ck2 = pkg.CloudKey(gpu_params(pkg, p), bsk_torus=bsk_t, ksk=ksk)
    ck2.ctx.set_option("quad_max", 0)
    ck4 = pkg.CloudKey(gpu_params(pkg, p), bsk_torus=bsk_t, ksk=ksk)
    ck4.ctx.set_option("quad_max", 1000000)
    # ... and with the eight-wave kernel switched off (it serves L = 3 launches of at most one bootstrap per CU)
    ck4only = None
    if which != "uint3":                                                   # L = 1: no eight-wave form
        ck4only = pkg.CloudKey(gpu_params(pkg, p), bsk_torus=bsk_t, ksk=ksk)
        ck4only.ctx.set_option("quad_max", 1000000)
        ck4only.ctx.set_option("oct_max", 0)
    rs = np.random.RandomState(21)
    for B in (1, 5, 300):            # 300 > one workgroup per CU: past the small-batch kernels, both contexts take the paired two-wave form
        cts = rand_u32(rs, (B, p.n + 1))
        cts[0] = 0
        tvs = rand_u32(rs, (B, 2, p.N))
        for nsteps in ((0, 1, 3, -1) if which == "small" else (0, 1)):
            a, b = ck2.ctx.blind_rotate_batch(cts, tvs, nsteps), ck4.ctx.blind_rotate_batch(cts, tvs, nsteps)
            if which == "small":
                assert np.array_equal(a, b), (B, nsteps)
                assert np.array_equal(a, ck4only.ctx.blind_rotate_batch(cts, tvs, nsteps)), (B, nsteps)
            else:
                # values reach 2^52 / 2^64 here: not exact integers for any fp64 pipeline, and one
                # differing low bit flips a digit of the next step, so only ONE product is comparable: the two
                # transform orders agree to within twice the per-product bound of test_gpu_uint5.py
                tol = 2 * (2**4 if which == "uint1" else 2**12)
                diff = (a.astype(np.int64) - b.astype(np.int64) + 2**31) % 2**32 - 2**31
                assert np.abs(diff).max() <= tol, (which, B, nsteps, np.abs(diff).max())
                if ck4only is not None:
                    c = ck4only.ctx.blind_rotate_batch(cts, tvs, nsteps)
                    diff = (a.astype(np.int64) - c.astype(np.int64) + 2**31) % 2**32 - 2**31
                    assert np.abs(diff).max() <= tol, (which, B, nsteps, np.abs(diff).max())
    ck2.close(); ck4.close()
    if ck4only is not None:
        ck4only.close()


def test_committed_golden_vectors_on_gpu(pkg, oracle):
    # tests/golden/*.npz (inputs + expected outputs, made by tests/golden/make_golden.py and checked there against the
    # exact-integer product) through the C ABI: one external product at the 128-bit ring/gadget, and a full
    # tiny-n bootstrap chain (accumulator after blind rotate, LWE after extract + key switch).
    import os
    from conftest import KeySet
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gdir, "extprod_N1024_L3_Bg6.npz"))
    p1 = oracle.params("128").small(1)
    zero_ksk = lambda p: np.zeros((p.N * p.t * (1 << p.basebit), p.n + 1), np.uint32)
    ck = pkg.CloudKey(gpu_params(pkg, p1), bsk_torus=np.ascontiguousarray(g["trgsw_torus"][None]), ksk=zero_ksk(p1))
    assert np.array_equal(ck.ctx.external_product_batch(0, g["trlwe_in"][None])[0], g["trlwe_out"])
    ck.close()
    g = np.load(os.path.join(gdir, "bootstrap_n6_seed7F4E0011.npz"))
    ks = KeySet(oracle, "128", int(g["seed"]), n_override=6)
    for kw in ({"bsk_fourier": ks.bsk}, {"bsk_torus": ks.bsk_torus}):
        ck = pkg.CloudKey(gpu_params(pkg, ks.p), ksk=ks.ksk, **kw)
        assert np.array_equal(ck.ctx.blind_rotate_batch(g["lwe_in"]), g["trlwe_acc"])
        assert np.array_equal(ck.ctx.extract_keyswitch_batch(g["trlwe_acc"]), g["lwe_out"])
        assert np.array_equal(ck.ctx.bootstrap_batch(g["lwe_in"]), g["lwe_out"])
        ck.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_batches_every_entry_point_agrees(oracle, keys_small, ck_small, pkg, seed):
    # Randomised cross-check of the dispatch machinery: a random batch size (any launcher range), random per-item
    # gates incl. MUX.  The host-pointer call, the device-pointer call and the same items issued in random pieces give
    # the same words; a sample is compared with the oracle.
    import torch
    k = keys_small
    rs = np.random.RandomState(7000 + seed)
    B = int(rs.choice([rs.randint(1, 40), rs.randint(200, 300), rs.randint(500, 800), rs.randint(1000, 1100), rs.randint(1500, 2600)]))
    a, b, c = (rand_u32(rs, (B, k.p.n + 1)) for _ in range(3))
    pool = [x for x in pkg.OPS if x != "MUX"] + ["MUX"] * 3
    ops = np.array([pkg.OPS[pool[i]] for i in rs.randint(0, len(pool), B)], np.uint8)
    host = ck_small.ctx.gate_batch(ops, a, b, c)
    ad, bd, cdv = (torch.from_numpy(v.view(np.int32)).cuda() for v in (a, b, c))
    od = torch.empty_like(ad)
    ck_small.ctx.gate_batch_dev(torch.from_numpy(ops).cuda(), ad, bd, cdv, od)
    torch.cuda.synchronize()
    assert np.array_equal(od.cpu().numpy().view(np.uint32), host), (seed, B)
    cuts = sorted(set([0, B] + list(rs.randint(0, B + 1, 3))))
    pieces = np.concatenate([ck_small.ctx.gate_batch(ops[i:j], a[i:j], b[i:j], c[i:j]) for i, j in zip(cuts[:-1], cuts[1:]) if j > i])
    assert np.array_equal(pieces, host), (seed, B, cuts)
    sample = sorted(set(rs.randint(0, B, 6)))
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops[sample], np.ascontiguousarray(a[sample]), np.ascontiguousarray(b[sample]),
                               np.ascontiguousarray(c[sample]))
    assert np.array_equal(host[sample], ref), (seed, B)


def test_host_pointer_batches_longer_than_a_slab(oracle, keys_small, ck_small, pkg):
    # tfhe_gate_batch on more than one pipeline piece (16,384 bootstraps) moves piece s+1's operands up and piece s-1's
    # results down while piece s computes (double-buffered staging, three streams).  Same words as the same items issued in
    # calls that each fit one piece (the un-pipelined path), with per-item ops incl. MUX, a ragged last piece, and twice in
    # a row (buffer reuse across calls).
    k = keys_small
    B = 2 * 16384 + 37
    rs = np.random.RandomState(91)
    a, b, c = (rand_u32(rs, (B, k.p.n + 1)) for _ in range(3))
    ops = np.array([pkg.OPS[x] for x in ("NAND", "XOR", "MUX", "OR")], np.uint8)[rs.randint(0, 4, B)]
    whole = ck_small.ctx.gate_batch(ops, a, b, c)
    pieces = np.concatenate([ck_small.ctx.gate_batch(ops[i:i + 9000], a[i:i + 9000], b[i:i + 9000], c[i:i + 9000])
                             for i in range(0, B, 9000)])
    assert np.array_equal(whole, pieces)
    assert np.array_equal(ck_small.ctx.gate_batch(ops, a, b, c), whole)
    sample = [0, 16383, 16384, 32767, 32768, B - 1]
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops[sample], np.ascontiguousarray(a[sample]), np.ascontiguousarray(b[sample]),
                               np.ascontiguousarray(c[sample]))
    assert np.array_equal(whole[sample], ref)
    uni = ck_small.ctx.gate_batch("NAND", a, b)
    sel = np.where(ops == pkg.OPS["NAND"])[0]
    assert np.array_equal(uni[sel], whole[sel])


def test_concurrent_host_threads(oracle, keys_small, ck_small, pkg):
    # A Go shim calls from many goroutines (the reference fans batches out over goroutines, trgsw.go:234-252).
    # ctypes drops the GIL, so these threads really overlap: four on one shared context (serialised by its
    # mutex), two on a second context of the same device.  Every result equals the single-threaded one.
    import threading
    k = keys_small
    ck2 = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    rs = np.random.RandomState(77)
    jobs = []
    for t in range(6):
        B = [40, 300, 7, 129, 64, 33][t]
        jobs.append((ck_small if t < 4 else ck2, ["NAND", "XOR", "OR", "AND", "NOR", "XNOR"][t],
                     rand_u32(rs, (B, k.p.n + 1)), rand_u32(rs, (B, k.p.n + 1))))
    want = [ck.ctx.gate_batch(op, a, b).copy() for ck, op, a, b in jobs]
    got, errs = [None] * len(jobs), []

    def run(i):
        try:
            ck, op, a, b = jobs[i]
            for _ in range(3):
                got[i] = ck.ctx.gate_batch(op, a, b).copy()
        except Exception as e:          # noqa: BLE001 - surfaced below
            errs.append((i, repr(e)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errs, errs
    for i in range(len(jobs)):
        assert np.array_equal(got[i], want[i]), i
    ck2.close()


def test_options_are_per_context_and_validated(pkg, keys_small):
    # tfhe_ctx_set_option / tfhe_ctx_get_option (include/tfhe_hip.h): defaults, round trip, negative = default, unknown
    # option refused; nothing comes from the environment (a stray TFHE_* variable must not change dispatch)
    import os
    os.environ["TFHE_QUAD_MAX"] = "0"
    os.environ["TFHE_KS_MFMA_MIN"] = "0"
    try:
        ck = pkg.CloudKey(gpu_params(pkg, keys_small.p), bsk_fourier=keys_small.bsk, ksk=keys_small.ksk)
    finally:
        del os.environ["TFHE_QUAD_MAX"], os.environ["TFHE_KS_MFMA_MIN"]
    ctx = ck.ctx
    cus = ctx.get_option("quad_max")
    assert cus > 0 and ctx.get_option("oct_max") == cus and ctx.get_option("ks_mfma_min") == 24 and ctx.get_option("frozen") == 0
    ctx.set_option("quad_max", 7); ctx.set_option("oct_max", 3); ctx.set_option("ks_mfma_min", 100)
    assert (ctx.get_option("quad_max"), ctx.get_option("oct_max"), ctx.get_option("ks_mfma_min")) == (7, 3, 100)
    for name in ("quad_max", "oct_max", "ks_mfma_min"):
        ctx.set_option(name, -1)
    assert (ctx.get_option("quad_max"), ctx.get_option("oct_max"), ctx.get_option("ks_mfma_min")) == (cus, cus, 24)
    with pytest.raises(pkg.TfheError, match="unknown option"):
        ctx._check(ctx._lib.tfhe_ctx_set_option(ctx._h, 99, 1))
    # the same gates under every dispatch setting: identical ciphertexts
    a, b = keys_small.enc([0, 1, 1, 0] * 4), keys_small.enc([1, 1, 0, 0] * 4)
    want = ctx.gate_batch("XOR", a, b)
    for q, o, m in ((0, 0, 0), (1000000, 0, 1), (1000000, 1000000, 0)):
        ctx.set_option("quad_max", q); ctx.set_option("oct_max", o); ctx.set_option("ks_mfma_min", m)
        assert np.array_equal(ctx.gate_batch("XOR", a, b), want), (q, o, m)
    ck.close()


def test_device_pointer_batch_longer_than_a_slab_with_mux(oracle, keys_small, ck_small, pkg):
    # tfhe_gate_batch_dev walks a batch in slabs of 65,536 items (fixed-size intermediate buffers); the MUX passes -- device-side
    # compaction, two extra launches, scatter -- run per slab.  One call across the slab boundary with per-item ops incl. MUX, ragged
    # second slab: the same words as the host-pointer path (which issues pieces of 16,384), and the oracle's on both sides of the boundary.
    import torch
    k = keys_small
    n1 = k.p.n + 1
    B = 65536 + 300
    rs = np.random.RandomState(92)
    a, b, c = (rand_u32(rs, (B, n1)) for _ in range(3))
    ops = np.array([pkg.OPS[x] for x in ("NAND", "XOR", "MUX", "ANDNY", "MUX")], np.uint8)[rs.randint(0, 5, B)]
    dev = lambda x: torch.from_numpy(x.view(np.int32)).cuda()
    d_out = torch.empty((B, n1), dtype=torch.int32, device="cuda")
    ck_small.ctx.gate_batch_dev(torch.from_numpy(ops).cuda(), dev(a), dev(b), dev(c), d_out)
    ck_small.ctx.sync()
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, ck_small.ctx.gate_batch(ops, a, b, c))
    sample = [0, 65535, 65536, 65537, B - 1] + [int(i) for i in np.where(ops == pkg.OPS["MUX"])[0][[0, -1]]]
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops[sample], np.ascontiguousarray(a[sample]), np.ascontiguousarray(b[sample]),
                               np.ascontiguousarray(c[sample]))
    assert np.array_equal(got[sample], ref)
