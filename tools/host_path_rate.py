#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (what a cgo caller gets): tfhe_gate_batch on numpy arrays,
pageable and page-locked, for a few batch sizes.  python tools/host_path_rate.py [B ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build(); pkg = g.load_package()
p = pkg.params.Security128Bit
rs = np.random.RandomState(1)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
for B in [int(x) for x in sys.argv[1:]] or [1024, 16384, 65536]:
    a, b = rnd((B, p.n + 1)), rnd((B, p.n + 1))
    ck.ctx.gate_batch("NAND", a, b)
    t = time.perf_counter(); reps = 3
    for _ in range(reps): ck.ctx.gate_batch("NAND", a, b)
    dt = (time.perf_counter() - t) / reps
    pa, pb, po = pkg.PinnedArray(a.shape), pkg.PinnedArray(a.shape), pkg.PinnedArray(a.shape)
    pa.array[...] = a; pb.array[...] = b
    ck.ctx.gate_batch("NAND", pa.array, pb.array, out=po.array)
    t = time.perf_counter()
    for _ in range(reps): ck.ctx.gate_batch("NAND", pa.array, pb.array, out=po.array)
    dtp = (time.perf_counter() - t) / reps
    print(f"B={B:6d}  pageable {dt*1e3:8.2f} ms = {B/dt:9.0f} gates/s   page-locked {dtp*1e3:8.2f} ms = {B/dtp:9.0f} gates/s", flush=True)
