import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import __graft_entry__ as g
g.build(); pkg = g.load_package()
from oracle_lib import Oracle
from conftest import KeySet, gpu_params
from go_tfhe_amd.circuits import ripple_carry_adder, CircuitExecutor
o = Oracle(); k = KeySet(o, "128", 0x7F4E0002, torus=False)
ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
C, bits = 256, 8
levels, nw, sums, cout = ripple_carry_adder(bits)
rs = np.random.RandomState(41)
av, bv = rs.randint(0, 256, C), rs.randint(0, 256, C)
n1 = k.p.n + 1
wires = np.zeros((nw, C, n1), np.uint32)
plain = np.zeros((nw, C), bool)
for i in range(bits):
    wires[i] = k.enc((av >> i) & 1); wires[bits + i] = k.enc((bv >> i) & 1)
    plain[i] = (av >> i) & 1; plain[bits + i] = (bv >> i) & 1
wt = torch.from_numpy(wires.view(np.int32)).cuda()
f = {"XOR": np.logical_xor, "AND": np.logical_and, "OR": np.logical_or}
for li, lvl in enumerate(levels):
    ex = CircuitExecutor(ck.ctx, [lvl], nw)
    ex.run(wt); torch.cuda.synchronize()
    res = wt.cpu().numpy().view(np.uint32)
    new = {out: f[op](plain[x], plain[y]) for op, x, y, z, out in lvl}
    for kk, v in new.items(): plain[kk] = v
    for op, x, y, z, out in lvl:
        d = k.dec(res[out])
        bad = (d != plain[out]).sum()
        if bad: print("level", li, op, x, y, "->", out, "wrong", bad, "of", C)
print("done")
# --- diagnose level 2 explicitly
wt = torch.from_numpy(wires.view(np.int32)).cuda()
for lvl in levels[:2]:
    CircuitExecutor(ck.ctx, [lvl], nw).run(wt)
torch.cuda.synchronize()
res = wt.cpu().numpy().view(np.uint32).copy()
x, y = np.ascontiguousarray(res[34]), np.ascontiguousarray(res[41])
ph = lambda arr: np.array([o.phase(k.p, k.s0, np.ascontiguousarray(r)) for r in arr]).astype(np.int64)
px, py = ph(x), ph(y)
print("phase/2^29 of in0:", np.round((px[:8] if True else 0) / 2**29, 3), "in1:", np.round(py[:8] / 2**29, 3))
h = ck.ctx.gate_batch("OR", x, y)
print("host-path OR ok:", (k.dec(h) == (k.dec(x) | k.dec(y))).all())
tx = wt[34].contiguous(); ty = wt[41].contiguous(); to = torch.empty_like(tx)
ck.ctx.gate_batch_dev("OR", tx, ty, None, to); torch.cuda.synchronize()
print("dev-path OR ok:", (k.dec(to.cpu().numpy().view(np.uint32)) == (k.dec(x) | k.dec(y))).all())
ex = CircuitExecutor(ck.ctx, [levels[2]], nw); ex.run(wt); torch.cuda.synchronize()
r2 = wt.cpu().numpy().view(np.uint32)
print("executor OR ok:", (k.dec(r2[42]) == (k.dec(x) | k.dec(y))).all(), "inputs unchanged:", np.array_equal(r2[34], x), np.array_equal(r2[41], y))
print("executor out == host out:", np.array_equal(r2[42], h))
