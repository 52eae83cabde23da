#!/usr/bin/env python3
"""Two-wave vs four-wave blind rotate (kernels.hpp vs kernels_quad.hpp) on ONE box, interleaved:
   python tools/quad_sweep.py [--rounds 4] [--launches 6] [--sizes 1,64,128,256,384,512]
Two contexts on the same random 128-bit key, one with option quad_max = 0 (always two waves per
bootstrap), one with quad_max = <huge>; checks the two kernels' accumulators are bit-identical and prints
the blind-rotate kernel time per batch size (HIP events, tfhe_last_kernel_ms)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--sizes", default="1,64,128,256,384,512")
ap.add_argument("--params", default="128")
ap.add_argument("--quad-limit", type=int, default=1 << 20)
args = ap.parse_args()
g.build(); pkg = g.load_package()
p = pkg.params.BY_NAME[args.params]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
bsk, ksk = rnd((p.n, 2 * p.L, 2, p.N)), rnd((p.ksk_rows, p.n + 1))
ck2 = pkg.CloudKey(p, bsk_torus=bsk, ksk=ksk); ck2.ctx.set_option("quad_max", 0)
ck4 = pkg.CloudKey(p, bsk_torus=bsk, ksk=ksk); ck4.ctx.set_option("quad_max", args.quad_limit)
sizes = [int(x) for x in args.sizes.split(",")]
Bmax = max(sizes)
cts = torch.from_numpy(rnd((Bmax, p.n + 1)).view(np.int32)).cuda()
res = {}
for B in sizes:
    c = cts[:B].contiguous()
    o2 = torch.empty((B, 2, p.N), dtype=torch.int32, device="cuda"); o4 = torch.empty_like(o2)
    ck2.ctx.blind_rotate_batch_dev(c, None, o2); ck4.ctx.blind_rotate_batch_dev(c, None, o4)
    torch.cuda.synchronize()
    same = bool(torch.equal(o2, o4))
    t2, t4 = [], []
    for _ in range(args.rounds):
        for ck, o, t in ((ck2, o2, t2), (ck4, o4, t4)):
            for _ in range(args.launches):
                ck.ctx.blind_rotate_batch_dev(c, None, o); torch.cuda.synchronize()
                t.append(ck.ctx.last_kernel_ms(0))
    res[B] = {"identical": same, "two_wave_ms": float(np.mean(t2)), "two_wave_min": float(np.min(t2)),
              "four_wave_ms": float(np.mean(t4)), "four_wave_min": float(np.min(t4)), "n": len(t2)}
    print(f"B={B:5d} identical={same}  two-wave {np.mean(t2):.3f} ms (min {np.min(t2):.3f})   four-wave {np.mean(t4):.3f} ms (min {np.min(t4):.3f})",
          flush=True)
print(json.dumps(res))
