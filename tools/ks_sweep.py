#!/usr/bin/env python3
"""Vector-ALU vs matrix-core key switch (kernels.hpp vs keyswitch_mfma.hpp) on ONE box, interleaved:
   python tools/ks_sweep.py [--sizes 16,64,128,256,512,1024,4096] [--params 128]
Two contexts on the same random key, one with option ks_mfma_min = 0 (never), one with ks_mfma_min = 1
(always); checks the outputs are bit-identical and prints the key-switch time per batch size (HIP events)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--sizes", default="16,64,128,256,512,1024,4096")
ap.add_argument("--params", default="128")
args = ap.parse_args()
g.build(); pkg = g.load_package()
p = pkg.params.BY_NAME[args.params]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
bsk, ksk = rnd((p.n, 2 * p.L, 2, p.N)), rnd((p.ksk_rows, p.n + 1))
def make(v):
    ck = pkg.CloudKey(p, bsk_torus=bsk, ksk=ksk)
    ck.ctx.set_option("ks_mfma_min", v)
    return ck
ckv, ckm = make(0), make(1)
sizes = [int(x) for x in args.sizes.split(",")]
trl = torch.from_numpy(rnd((max(sizes), 2, p.N)).view(np.int32)).cuda()
res = {}
for B in sizes:
    x = trl[:B].contiguous()
    ov = torch.empty((B, p.n + 1), dtype=torch.int32, device="cuda"); om = torch.empty_like(ov)
    ckv.ctx.extract_keyswitch_batch_dev(x, ov); ckm.ctx.extract_keyswitch_batch_dev(x, om)
    torch.cuda.synchronize()
    same = bool(torch.equal(ov, om))
    tv, tm = [], []
    for _ in range(args.rounds):
        for ck, o, t in ((ckv, ov, tv), (ckm, om, tm)):
            for _ in range(args.launches):
                ck.ctx.extract_keyswitch_batch_dev(x, o); torch.cuda.synchronize()
                t.append(ck.ctx.last_kernel_ms(1))
    res[B] = {"identical": same, "vector_ms": float(np.mean(tv)), "mfma_ms": float(np.mean(tm))}
    print(f"B={B:5d} identical={same}  vector {np.mean(tv):.3f} ms (min {np.min(tv):.3f})   matrix-core {np.mean(tm):.3f} ms (min {np.min(tm):.3f})", flush=True)
print(json.dumps(res))
