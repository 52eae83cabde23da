import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import __graft_entry__ as g
g.build(); pkg = g.load_package()
from oracle_lib import Oracle
from conftest import KeySet, gpu_params
o = Oracle(); k = KeySet(o, "128", 0x7F4E0002, torus=False)
ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
rs = np.random.RandomState(1)
for B in (4, 256):
    A = rs.randint(0, 2, B); Bb = rs.randint(0, 2, B)
    a, b = k.enc(A), k.enc(Bb)
    for op, f in (("OR", np.logical_or), ("AND", np.logical_and), ("NAND", lambda x, y: ~(x & y))):
        h = ck.ctx.gate_batch(op, a, b)
        ta = torch.from_numpy(a.view(np.int32)).cuda(); tb = torch.from_numpy(b.view(np.int32)).cuda(); to = torch.empty_like(ta)
        ck.ctx.gate_batch_dev(op, ta, tb, None, to); torch.cuda.synchronize()
        d = to.cpu().numpy().view(np.uint32)
        want = f(A.astype(bool), Bb.astype(bool))
        print(B, op, "host ok:", (k.dec(h) == want).all(), "dev ok:", (k.dec(d) == want).all(), "dev==host:", np.array_equal(d, h))
    # second-generation inputs: AND then OR of results
    x = ck.ctx.gate_batch("AND", a, b); y = ck.ctx.gate_batch("XOR", a, b)
    z = ck.ctx.gate_batch("OR", x, y)
    print(B, "OR(AND,XOR) host ok:", (k.dec(z) == ((A & Bb) | (A ^ Bb)).astype(bool)).all())
