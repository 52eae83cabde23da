"""GPU tier, boxes with MORE THAN ONE visible device (a multi-GPU node, or one MI355X in DPX / CPX compute-partition mode -- which is
how tools/cpx_functional.sh runs this file on the one-GPU box): the code of SURVEY.md 8(e) that one device cannot execute.

  * tfhe_ctx_clone_to between two DIFFERENT devices (include/tfhe_hip.h; the peer-copy branch, hipMemcpyPeerAsync, or the documented
    host-staged fallback when the two are not peers) -- replica indistinguishable from its source;
  * CloudKeySet over every visible device from one process (trgsw.BatchBlindRotate's fan-out, trgsw/trgsw.go:234-252);
  * RCCL with N > 1 ranks: key broadcast, packed scatter / gather of a ragged mixed batch, circuits sharded by circuit
    (tests/multirank_worker.py, one process per device, world 2 and world = all devices up to 8).

Skipped (with the device count in the reason) on a one-device box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import rand_u32

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
needs2 = pytest.mark.skipif(NDEV < 2, reason=f"{NDEV} device(s) visible: the cross-device paths need two (multi-GPU node or DPX/CPX partition mode)")


@needs2
def test_clone_to_another_device_is_indistinguishable_from_its_source(pkg, keys_small, ck_small):
    k = keys_small
    rep = ck_small.clone_to(1)
    try:
        path = rep.ctx.get_option("clone_path")
        assert path in (2, 3), path                               # 2 = peer copy (hipMemcpyPeerAsync), 3 = staged through the host: never the same-device path
        print(f"clone_path device 0 -> 1: {path} ({'peer' if path == 2 else 'host-staged'})")
        for which in (0, 1):
            assert torch.equal(rep.ctx.key_export_dev(which).cpu(), ck_small.ctx.key_export_dev(which).cpu())
        rs = np.random.RandomState(21)
        n1 = k.p.n + 1
        a, b, c = (rand_u32(rs, (37, n1)) for _ in range(3))
        ops = rs.randint(0, 11, size=37).astype(np.uint8)
        assert np.array_equal(rep.ctx.gate_batch(ops, a, b, c), ck_small.ctx.gate_batch(ops, a, b, c))
        back = rep.clone_to(0)                                    # and back again: a clone of a clone on the first device
        try:
            assert np.array_equal(back.ctx.gate_batch("NAND", a, b), ck_small.ctx.gate_batch("NAND", a, b))
        finally:
            back.close()
    finally:
        rep.close()


@needs2
def test_cloud_key_set_over_every_device_equals_one_device(pkg, oracle, keys_small, ck_small):
    k = keys_small
    ks = pkg.CloudKeySet(ck_small, list(range(NDEV)))
    try:
        assert len(ks) == NDEV
        rs = np.random.RandomState(22)
        n1 = k.p.n + 1
        B = 5 * NDEV + 3
        a, b, c = (rand_u32(rs, (B, n1)) for _ in range(3))
        ops = rs.randint(0, 11, size=B).astype(np.uint8)
        got = ks.gate_batch(ops, a, b, c)
        want, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops, a, b, c)
        assert np.array_equal(got, want)
        assert np.array_equal(got, ck_small.ctx.gate_batch(ops, a, b, c))
    finally:
        ks.close()


def run_world(world, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    port = 29700 + (os.getpid() + world) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "multirank_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, f"world {world}: rc {r.returncode}\nstdout tail: {r.stdout[-1500:]}\nstderr tail: {r.stderr[-3000:]}"
    return json.loads(lines[-1])


def test_the_multirank_worker_itself_runs_over_rccl_with_one_rank(built):
    # the worker the multi-device tests launch, executed where only one device exists: one RCCL rank goes through the key broadcast,
    # the packed scatter / gather, the circuits sharded by circuit and the per-rank oracle comparison -- so that the first box with
    # two devices tests the fan-out, not the script
    rec = run_world(1)
    assert rec["world_size"] == 1 and rec["collective_backend"] == "nccl" and rec["ranks_verified"] == 1 and rec["verified"]
    assert all(rec["root_checks"].values()), rec["root_checks"]


@needs2
@pytest.mark.parametrize("world", sorted({2, min(NDEV, 8)} if NDEV >= 2 else {2}))
def test_rccl_sharded_gates_and_circuits_with_several_ranks(built, world):
    rec = run_world(world)
    print(json.dumps(rec))
    assert rec["world_size"] == world and rec["collective_backend"] == "nccl"
    assert rec["ranks_verified"] == world and rec["verified"]
    assert all(rec["root_checks"].values()), rec["root_checks"]
    assert sorted(r["device"] for r in rec["per_rank"]) == list(range(world))
