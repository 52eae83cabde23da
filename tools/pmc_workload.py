#!/usr/bin/env python3
"""Workload for the PMC passes at any parameter set (random keys; timing is value-independent):
   python tools/pmc_workload.py <params name> <batch> [launches]
Gate sets run NAND gates, Uint sets a LUT bootstrap -- the same two kernels either way."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
pname, B = sys.argv[1], int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
p = pkg.params.BY_NAME[pname]
rs = np.random.RandomState(1)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
a = torch.from_numpy(rnd((B, p.n + 1)).view(np.int32)).cuda()
b = torch.from_numpy(rnd((B, p.n + 1)).view(np.int32)).cuda()
lut = torch.from_numpy(rnd((2, p.N)).view(np.int32)).cuda()
out = torch.empty_like(a)
for _ in range(L):
    if pname in ("80", "110", "128"):
        ck.ctx.gate_batch_dev("NAND", a, b, None, out)
    else:
        ck.ctx.bootstrap_batch_dev(a, lut, out)
torch.cuda.synchronize()
print("BR ms", ck.ctx.last_kernel_ms(0), "KS ms", ck.ctx.last_kernel_ms(1))
