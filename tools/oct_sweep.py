#!/usr/bin/env python3
"""Four-wave vs eight-wave blind rotate (kernels_quad.hpp) on ONE box, interleaved:
   python tools/oct_sweep.py [--rounds 4] [--launches 6] [--sizes 1,32,64,128,256]
Two contexts on the same random 128-bit key, one with option oct_max = 0 (four waves per bootstrap for
launches of at most one bootstrap per CU), one with the default; checks the accumulators are bit-identical (also
against the two-wave kernel) and prints the blind-rotate kernel time per batch size (HIP events)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--sizes", default="1,32,64,128,256")
args = ap.parse_args()
g.build(); pkg = g.load_package()
p = pkg.params.BY_NAME["128"]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
bsk, ksk = rnd((p.n, 2 * p.L, 2, p.N)), rnd((p.ksk_rows, p.n + 1))
def make(opts):
    ck = pkg.CloudKey(p, bsk_torus=bsk, ksk=ksk)
    for k, v in opts.items(): ck.ctx.set_option(k, v)
    return ck
ck2 = make({"quad_max": 0})
ck4 = make({"oct_max": 0})
ck8 = make({})
sizes = [int(x) for x in args.sizes.split(",")]
cts = torch.from_numpy(rnd((max(sizes), p.n + 1)).view(np.int32)).cuda()
res = {}
for B in sizes:
    c = cts[:B].contiguous()
    outs = [torch.empty((B, 2, p.N), dtype=torch.int32, device="cuda") for _ in range(3)]
    for ck, o in zip((ck2, ck4, ck8), outs): ck.ctx.blind_rotate_batch_dev(c, None, o)
    torch.cuda.synchronize()
    same = bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))
    t4, t8 = [], []
    for _ in range(args.rounds):
        for ck, o, t in ((ck4, outs[1], t4), (ck8, outs[2], t8)):
            for _ in range(args.launches):
                ck.ctx.blind_rotate_batch_dev(c, None, o); torch.cuda.synchronize()
                t.append(ck.ctx.last_kernel_ms(0))
    res[B] = {"identical": same, "four_wave_ms": float(np.mean(t4)), "eight_wave_ms": float(np.mean(t8)), "n": len(t4)}
    print(f"B={B:5d} identical={same}  four-wave {np.mean(t4):.3f} ms (min {np.min(t4):.3f})   eight-wave {np.mean(t8):.3f} ms (min {np.min(t8):.3f})", flush=True)
print(json.dumps(res))
