"""GPU tier: the batch scatter / compute / gather path over RCCL (backend "nccl") on the GPUs this
box has.  With one GPU the group has one rank, which still drives ShardedGates + gpu_compute
(device-pointer ABI on torch's stream) through the NCCL collectives; the 2-rank logic is covered
by tests/test_distributed_cpu.py (gloo) and the 8-GPU run belongs to the driver's bench."""
import os

import numpy as np
import pytest
import torch

from conftest import rand_u32

pytestmark = pytest.mark.gpu


def test_sharded_gates_nccl_single_node(oracle, keys_small, ck_small, pkg):
    import torch.distributed as dist
    from go_tfhe_amd.distributed import ShardedGates, gpu_compute
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        k = keys_small
        n1 = k.p.n + 1
        eng = ShardedGates(gpu_compute(ck_small.ctx), n1, device=dev)
        rs = np.random.RandomState(51)
        B = 37
        a, b, c = (rand_u32(rs, (B, n1)) for _ in range(3))
        names = np.array(["AND", "OR", "XOR", "MUX", "NAND"])[rs.randint(0, 5, B)]
        ops = np.array([pkg.OPS[x] for x in names], np.uint8)
        ta, tb, tc = (torch.from_numpy(x.view(np.int32)).to(dev) for x in (a, b, c))
        got = eng.gate_batch(torch.from_numpy(ops).to(dev), ta, tb, tc)
        torch.cuda.synchronize()
        want, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops, a, b, c)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want)
        got2 = eng.gate_batch("XNOR", ta, tb)
        want2, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, "XNOR", a, b)
        assert np.array_equal(got2.cpu().numpy().view(np.uint32), want2)
        assert set(eng.last_timing) == {"scatter_s", "compute_s", "gather_s"}
        # key replication: the broadcast helper (a one-rank broadcast here) and the blob round trip it is made of --
        # a second context that only ever IMPORTS the two device-layout blobs computes identical ciphertexts
        from go_tfhe_amd.distributed import broadcast_cloud_key
        from conftest import gpu_params
        broadcast_cloud_key(ck_small.ctx, src=0)
        ck2 = pkg.CloudKey(gpu_params(pkg, k.p))
        with pytest.raises(pkg.TfheError):
            ck2.ctx.gate_batch("NAND", a, b)                      # no key yet
        for which in (0, 1):
            blob = ck_small.ctx.key_export_dev(which)
            assert blob.numel() == ck_small.ctx.key_size(which)
            ck2.ctx.key_import_dev(which, blob)
        torch.cuda.synchronize()
        for B in (37, 300):                                       # both blind-rotate layouts of the imported key
            idx = np.arange(B) % 37
            assert np.array_equal(ck2.ctx.gate_batch("XNOR", a[idx], b[idx]), want2[idx])
        ck2.close()
    finally:
        dist.destroy_process_group()
