"""GPU tier: tfhe_ctx_clone_to -- one cloud key on several GPUs of a node from ONE process (the reference's
batch fan-out shares read-only keys across goroutines, trgsw.go:234-252).  The replica's keys are copied GPU to GPU behind
the C ABI (hipMemcpyPeerAsync; on the one GPU of the test box the peer copy degenerates to a device-to-device copy), and a
replica must be indistinguishable from its source: same key blobs, same output words."""
import numpy as np
import pytest
import torch

from conftest import KeySet, gpu_params, rand_u32

pytestmark = pytest.mark.gpu


def test_clone_has_the_same_key_blobs_and_the_same_outputs(pkg, keys_small, ck_small):
    k = keys_small
    rep = ck_small.clone_to(0)
    try:
        assert rep.ctx.get_option("clone_path") == 1            # same GPU: device-to-device
        assert ck_small.ctx.get_option("clone_path") == 0        # the source is not a clone
        for which in (0, 1):
            assert torch.equal(rep.ctx.key_export_dev(which).cpu(), ck_small.ctx.key_export_dev(which).cpu())
        rs = np.random.RandomState(11)
        n1 = k.p.n + 1
        a, b, c = (rand_u32(rs, (37, n1)) for _ in range(3))
        ops = rs.randint(0, 11, size=37).astype(np.uint8)       # all ten gates + MUX
        assert np.array_equal(rep.ctx.gate_batch(ops, a, b, c), ck_small.ctx.gate_batch(ops, a, b, c))
        # ... and a clone of the clone
        rep2 = rep.clone_to(0)
        assert np.array_equal(rep2.ctx.gate_batch("NAND", a, b), ck_small.ctx.gate_batch("NAND", a, b))
        rep2.close()
    finally:
        rep.close()
    # the source survives its replicas
    assert ck_small.ctx.gate_batch("AND", a[:2], b[:2]).shape == (2, n1)


def test_clone_carries_exactly_the_keys_the_source_holds_and_its_limits(pkg, keys_small):
    k = keys_small
    src = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk)            # bootstrapping key only
    src.ctx.set_option("oct_max", 0)
    src.ctx.set_option("combine_max", 7)
    rep = src.clone_to(0)
    try:
        assert rep.ctx.get_option("oct_max") == 0 and rep.ctx.get_option("combine_max") == 7
        a = rand_u32(np.random.RandomState(2), (3, k.p.n + 1))
        assert np.array_equal(rep.ctx.blind_rotate_batch(a), src.ctx.blind_rotate_batch(a))
        with pytest.raises(pkg.TfheError, match="not loaded"):
            rep.ctx.gate_batch("NAND", a, a)
        with pytest.raises(pkg.TfheError, match="not present"):
            src.clone_to(4096)
    finally:
        rep.close()
        src.close()


def test_clone_at_the_uint5_shape_bootstraps_identically(oracle, pkg):
    k = KeySet(oracle, "uint5", 0x7F4E0081, n_override=24, torus=False)
    src = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    rep = src.clone_to(0)
    try:
        lut = oracle.lut_generate(k.p, [(3 * x + 1) % 32 for x in range(32)])
        cts = np.stack([oracle.encrypt_message(k.p, k.rng, m, 32, k.s0) for m in range(0, 32, 3)])
        got, want = rep.ctx.bootstrap_batch(cts, lut), src.ctx.bootstrap_batch(cts, lut)
        assert np.array_equal(got, want)
        dec = [oracle.decrypt_message(k.p, 32, k.s0, row) for row in got]
        assert dec == [(3 * m + 1) % 32 for m in range(0, 32, 3)]
    finally:
        rep.close()
        src.close()


def test_cloud_key_set_shards_a_ragged_batch_in_order(pkg, keys_small, ck_small):
    k = keys_small
    ks = pkg.CloudKeySet(ck_small, [0, 0, 0])                    # three independent submitters on the one GPU
    try:
        assert len(ks) == 3 and ks.shards(7) == [(0, 2), (2, 4), (4, 7)]
        rs = np.random.RandomState(12)
        n1 = k.p.n + 1
        a, b, c = (rand_u32(rs, (7, n1)) for _ in range(3))
        ops = np.array([0, 10, 3, 4, 10, 9, 1], np.uint8)
        assert np.array_equal(ks.gate_batch(ops, a, b, c), ck_small.ctx.gate_batch(ops, a, b, c))
        assert np.array_equal(ks.gate_batch("XNOR", a[:2], b[:2]), ck_small.ctx.gate_batch("XNOR", a[:2], b[:2]))   # fewer items than replicas
    finally:
        ks.close()


def test_host_staged_fallback_path_gives_the_same_replica(pkg, keys_small, ck_small):
    # the documented fallback of tfhe_ctx_clone_to when the two devices are not peers (copy staged through 32 MB of page-locked host
    # memory), forced here on the one GPU of the box: same blobs, same outputs, clone path 3
    k = keys_small
    ck_small.ctx.set_option("clone_force_host", 1)
    try:
        rep = ck_small.clone_to(0)
    finally:
        ck_small.ctx.set_option("clone_force_host", 0)
    try:
        assert rep.ctx.get_option("clone_path") == 3
        for which in (0, 1):
            assert torch.equal(rep.ctx.key_export_dev(which).cpu(), ck_small.ctx.key_export_dev(which).cpu())
        rs = np.random.RandomState(13)
        a, b = rand_u32(rs, (9, k.p.n + 1)), rand_u32(rs, (9, k.p.n + 1))
        assert np.array_equal(rep.ctx.gate_batch("XOR", a, b), ck_small.ctx.gate_batch("XOR", a, b))
    finally:
        rep.close()
    assert ck_small.clone_to(0).ctx.get_option("clone_path") == 1          # back to the device-to-device path
