#!/usr/bin/env python3
"""Extended-table bootstrap (Uint6: polyExtendFactor 2) against the plain Uint5 bootstrap at the same batch, on ONE box:
   python tools/ext_bench.py [--batch 64] [--launches 8]
Random key and inputs (values do not affect timing).  ext = 2 runs the persistent eight-wave kernel, ext = 4 the
launch-per-step path; prints blind-rotate milliseconds per batch."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--launches", type=int, default=8)
args = ap.parse_args()
pkg = g.load_package()
p = pkg.params.BY_NAME["uint5"]
rs = np.random.RandomState(5)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
B = args.batch
cts = torch.from_numpy(rnd((B, p.n + 1)).view(np.int32)).cuda()
out = torch.empty_like(cts)
res = {}
for ext in (1, 2, 4):
    lut = torch.from_numpy(rnd((ext, 2, p.N)).view(np.int32)).cuda()
    fn = (lambda: ck.ctx.bootstrap_batch_dev(cts, lut[0], out)) if ext == 1 else (lambda: ck.ctx.bootstrap_extended_batch_dev(cts, lut, out))
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.launches):
        ck.ctx.timing_enable(True); fn(); torch.cuda.synchronize(); ck.ctx.timing_enable(False)
        n, ms = ck.ctx.timing_read(0); ck.ctx.timing_read(1)
        ts.append(ms)
    res[f"ext{ext}"] = {"blind_rotate_ms": float(np.mean(ts)), "min": float(np.min(ts))}
    print(f"batch {B} ext {ext}: blind rotate {np.mean(ts):.3f} ms (min {np.min(ts):.3f})", flush=True)
print(json.dumps(res))
