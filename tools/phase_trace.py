#!/usr/bin/env python3
"""Where a CMUX step of the blind-rotate kernels spends its time: run against a -DPHASE_TRACE build
   tools/build_variant.sh trace -DPHASE_TRACE
   TFHE_HIP_LIB=go-tfhe_amd/lib/variants/trace.so python tools/phase_trace.py [--batch 1024]
Prints shader-clock cycles per step between the marks of PhaseClock (kernels.hpp), per wave of item 0 and averaged over
the launch.  Batches of <= one bootstrap per CU run the eight-wave kernel (kernels_quad.hpp; marks: 0 keys+decompose,
1 forward transforms, 2 products+hand-over stores, 3 barrier 1, 4 gather (group 1), 5 inverse+store (group 1),
6 barrier 2 (group 0: the whole wait), 7 update, 8 barrier 3); larger ones the two-wave kernel (0 decompose, 1 forward transforms
(+ level-0 key request), 2 products + key refills + hand-over stores, 3 barrier 1, 4 gather, 5 barrier 2,
6 inverse+round, 7 accumulate)."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--groups", type=int, default=2, help="wave groups of the small-batch kernel in the traced build (OCT_GROUPS_L3)")
ap.add_argument("--params", default="128", help="128 (N = 1024 kernels) or uint5 (the four-wave N = 2048 kernel)")
args = ap.parse_args()
pkg = g.load_package()
p = pkg.params.BY_NAME[args.params]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
B = args.batch
c = torch.from_numpy(rnd((B, p.n + 1)).view(np.int32)).cuda()
o = torch.zeros((B, 2, p.N), dtype=torch.int32, device="cuda")
for _ in range(3): ck.ctx.blind_rotate_batch_dev(c, None, o)
torch.cuda.synchronize()
n2048 = p.N == 2048
oct_kernel = not n2048 and B <= torch.cuda.get_device_properties(0).multi_processor_count
W = 4 if n2048 else 4 * args.groups if oct_kernel else 2
NM = 10 if n2048 else 9 if oct_kernel else 8
t = o.cpu().numpy().view(np.int64).reshape(B, -1)[:, :16 * W].reshape(B, W, 16)[:, :, :NM] / p.n
names = (["extract", "barrier1", "fwd+mac", "barrier2", "gather", "(unused)", "inv+send", "barrier3", "update", "barrier4"] if n2048 else
         ["keys+dec", "forward", "mac+store", "barrier1", "gather", "inv+store", "barrier2", "update", "barrier3"] if oct_kernel else
         ["decompose", "forward", "mac+keys", "barrier1", "gather", "barrier2", "inverse", "update"])
print("kernel ms", ck.ctx.last_kernel_ms(0), "N=2048 four-wave" if n2048 else f"{4 * args.groups}-wave" if oct_kernel else "two-wave")
print("wave  " + "".join(f"{n:>10s}" for n in names) + "     total")
for w in range(W):
    row = t[0, w]
    print(f"w{w}    " + "".join(f"{v:10.0f}" for v in row) + f"{row.sum():10.0f}")
m, sd = t.mean(axis=0), t.std(axis=0)
print("mean (std) over the launch's items:")
for w in range(W):
    print(f"w{w}    " + "".join(f"{v:10.0f}" for v in m[w]) + f"{m[w].sum():10.0f}")
    print("      " + "".join(f"{'(%d)' % v:>10s}" for v in sd[w]))
if not oct_kernel and not n2048 and B % 2 == 0:
    print("mean by position of the item in a two-item workgroup (launches of more than one bootstrap per CU):")
    for pos in range(2):
        for w in range(W):
            mm = t[pos::2, w].mean(axis=0)
            print(f"i{pos} w{w} " + "".join(f"{v:10.0f}" for v in mm) + f"{mm.sum():10.0f}")
