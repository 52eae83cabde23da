#!/usr/bin/env python3
"""Wide key switch (k_keyswitch_wide, bases 16-64) with 64 vs 128 ciphertexts per wave on ONE box, interleaved:
   python tools/ks_wide_sweep.py [--params uint5] [--sizes 256,512,1024,2048]
Two contexts on the same random key, option ks_wide_ct = 64 / 128; bit-identity of the outputs and the key-switch time (HIP events)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ap = argparse.ArgumentParser()
ap.add_argument("--params", default="uint5")
ap.add_argument("--sizes", default="256,512,1024,2048")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--launches", type=int, default=8)
args = ap.parse_args()
g.build(); pkg = g.load_package()
p = pkg.params.BY_NAME[args.params]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
bsk, ksk = rnd((p.n, 2 * p.L, 2, p.N)), rnd((p.ksk_rows, p.n + 1))
cks = {}
for ct in (64, 128):
    ck = pkg.CloudKey(p, bsk_torus=bsk, ksk=ksk)
    ck.ctx.set_option("ks_wide_ct", ct)
    cks[ct] = ck
res = {}
for B in [int(x) for x in args.sizes.split(",")]:
    tr = torch.from_numpy(rnd((B, 2, p.N)).view(np.int32)).cuda()
    outs = {ct: torch.zeros((B, p.n + 1), dtype=torch.int32, device="cuda") for ct in cks}
    for ct, ck in cks.items():
        ck.ctx.extract_keyswitch_batch_dev(tr, outs[ct])
    torch.cuda.synchronize()
    same = bool(torch.equal(outs[64], outs[128]))
    t = {64: [], 128: []}
    for _ in range(args.rounds):
        for ct, ck in cks.items():
            for _ in range(args.launches):
                ck.ctx.extract_keyswitch_batch_dev(tr, outs[ct]); torch.cuda.synchronize()
                t[ct].append(ck.ctx.last_kernel_ms(1))
    res[B] = {"identical": same, "ct64_ms": float(np.mean(t[64])), "ct128_ms": float(np.mean(t[128]))}
    print(f"B={B:5d} identical={same}  64/wave {np.mean(t[64]):.3f} ms (min {np.min(t[64]):.3f})   128/wave {np.mean(t[128]):.3f} ms (min {np.min(t[128]):.3f})", flush=True)
print(json.dumps(res))
