#!/usr/bin/env python3
"""Where a CMUX step of the eight-wave kernel spends its time: run against a -DOCT_TRACE build
   tools/build_variant.sh oct_trace -DOCT_TRACE
   TFHE_HIP_LIB=go-tfhe_amd/lib/variants/oct_trace.so python tools/oct_trace.py [--batch 64]
Prints, per wave of workgroup 0 and averaged over the launch, shader-clock cycles per step between the marks of
OctTrace (kernels_quad.hpp): 0 keys+decompose, 1 forward transforms, 2 products+hand-over stores, 3 barrier 1,
4 gather (group 1), 5 inverse+store (group 1), 6 barrier 2 (group 0: the whole wait), 7 update."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
pkg = g.load_package()
p = pkg.params.BY_NAME["128"]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
B = args.batch
c = torch.from_numpy(rnd((B, p.n + 1)).view(np.int32)).cuda()
o = torch.zeros((B, 2, p.N), dtype=torch.int32, device="cuda")
for _ in range(3): ck.ctx.blind_rotate_batch_dev(c, None, o)
torch.cuda.synchronize()
t = o.cpu().numpy().view(np.int64).reshape(B, -1)[:, :128].reshape(B, 8, 16)[:, :, :8] / p.n
names = ["keys+dec", "forward", "mac+store", "barrier1", "gather", "inv+store", "barrier2", "update"]
print("kernel ms", ck.ctx.last_kernel_ms(0))
print("wave  " + "".join(f"{n:>10s}" for n in names) + "     total")
for w in range(8):
    row = t[0, w]
    print(f"w{w} g{w >> 2} " + "".join(f"{v:10.0f}" for v in row) + f"{row.sum():10.0f}")
m = t.mean(axis=0)
print("mean over workgroups:")
for w in range(8):
    print(f"w{w} g{w >> 2} " + "".join(f"{v:10.0f}" for v in m[w]) + f"{m[w].sum():10.0f}")
