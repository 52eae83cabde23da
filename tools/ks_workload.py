import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); p = pkg.params.BY_NAME["128"]
rs = np.random.RandomState(3)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
B = int(sys.argv[1])
x = torch.from_numpy(rnd((B, 2, p.N)).view(np.int32)).cuda(); o = torch.empty((B, p.n + 1), dtype=torch.int32, device="cuda")
for _ in range(10): ck.ctx.extract_keyswitch_batch_dev(x, o)
torch.cuda.synchronize()
