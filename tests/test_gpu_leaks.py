"""GPU tier: contexts give back what they took.  A service creates and drops cloud keys for its tenants; tfhe_ctx_destroy must release the
keys, the derived key layouts, the grow-only intermediate buffers, the streams / events and the page-locked staging of the combiner --
measured as the device's free memory (hipMemGetInfo through torch) and the process's page-locked + resident host memory."""
import gc
import os

import numpy as np
import pytest

from conftest import gpu_params


def _rss_kb():
    with open(f"/proc/{os.getpid()}/status") as fh:
        for line in fh:
            if line.startswith("VmRSS:"):
                return int(line.split()[1])
    return 0


@pytest.mark.gpu
def test_create_use_clone_destroy_cycles_return_device_and_host_memory(pkg, oracle, keys_small):
    import torch
    k = keys_small
    P = gpu_params(pkg, k.p)
    rs = np.random.RandomState(3)
    a = rs.randint(0, 2**32, (700, k.p.n + 1), dtype=np.uint64).astype(np.uint32)

    def cycle():
        ck = pkg.CloudKey(P, bsk_fourier=k.bsk, ksk=k.ksk)
        out = ck.ctx.gate_batch("NAND", a, a[::-1].copy())            # grows the intermediate buffers, pins staging memory
        ck.ctx.bootstrap_batch(a[:5], k.tv)
        rep = ck.clone_to(0)                                          # a replica with its own buffers
        assert np.array_equal(rep.ctx.gate_batch("NAND", a[:9], a[:9][::-1].copy()), ck.ctx.gate_batch("NAND", a[:9], a[:9][::-1].copy()))
        blob = ck.ctx.key_export(0)
        other = pkg.Context(P)
        other.key_import(0, blob)
        other.close(); rep.close(); ck.close()
        return out

    for _ in range(3):                                               # warm-up: the HIP runtime and torch keep pools of their own
        cycle()
    gc.collect(); torch.cuda.synchronize()
    free0, rss0 = torch.cuda.mem_get_info()[0], _rss_kb()
    for _ in range(25):
        cycle()
    gc.collect(); torch.cuda.synchronize()
    free1, rss1 = torch.cuda.mem_get_info()[0], _rss_kb()
    # one cycle holds ~3 contexts x (keys 4 MB at this reduced dimension + buffers for 700 items ~ 12 MB): 25 leaked cycles would be > 1 GB
    assert free0 - free1 < 64 << 20, f"device memory not returned: {(free0 - free1) >> 20} MiB after 25 cycles"
    assert rss1 - rss0 < 256 << 10, f"host memory grows: {(rss1 - rss0) >> 10} MiB after 25 cycles"
