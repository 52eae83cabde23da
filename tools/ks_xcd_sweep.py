#!/usr/bin/env python3
"""Wide key switch (k_keyswitch_wide) with the partial sums of its coefficient ranges meeting in one copy PER XCD + a reduce kernel
(option ks_xcd_sum = 1, round 5) against 32-bit atomics straight into the output (= 0, round 4), on ONE box, interleaved:
   python tools/ks_xcd_sweep.py [--params uint5] [--sizes 64,256,512,1024,2048,4096]
Two contexts on the same random key; bit-identity of the outputs and the whole key-switch time (HIP events around memset + kernel + reduce)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ap = argparse.ArgumentParser()
ap.add_argument("--params", default="uint5")
ap.add_argument("--sizes", default="64,256,512,1024,2048,4096")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--launches", type=int, default=8)
args = ap.parse_args()
g.build(); pkg = g.load_package()
p = pkg.params.BY_NAME[args.params]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
bsk, ksk = rnd((p.n, 2 * p.L, 2, p.N)), rnd((p.ksk_rows, p.n + 1))
src = pkg.CloudKey(p, bsk_torus=bsk, ksk=ksk)
cks = {1: src, 0: src.clone_to(0)}
cks[0].ctx.set_option("ks_xcd_sum", 0)
assert cks[1].ctx.get_option("ks_xcd_sum") == 1
res = {}
for B in [int(x) for x in args.sizes.split(",")]:
    tr = torch.from_numpy(rnd((B, 2, p.N)).view(np.int32)).cuda()
    outs = {k: torch.zeros((B, p.n + 1), dtype=torch.int32, device="cuda") for k in cks}
    for k, ck in cks.items():
        ck.ctx.extract_keyswitch_batch_dev(tr, outs[k])
    torch.cuda.synchronize()
    same = bool(torch.equal(outs[0], outs[1]))
    t = {0: [], 1: []}
    for _ in range(args.rounds):
        for k, ck in cks.items():
            for _ in range(args.launches):
                ck.ctx.extract_keyswitch_batch_dev(tr, outs[k]); torch.cuda.synchronize()
                t[k].append(ck.ctx.last_kernel_ms(1))
    res[B] = {"identical": same, "atomics_into_out_ms": float(np.mean(t[0])), "per_xcd_sum_ms": float(np.mean(t[1]))}
    print(f"B={B:5d} identical={same}  atomics into out {np.mean(t[0]):.3f} ms (min {np.min(t[0]):.3f})   per-XCD partial sums + reduce {np.mean(t[1]):.3f} ms (min {np.min(t[1]):.3f})", flush=True)
print(json.dumps(res))
