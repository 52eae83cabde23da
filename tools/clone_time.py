"""tools/clone_time.py (GPU box): what tfhe_ctx_clone_to costs on one GPU -- the device-to-device path and the forced host-staged path (the\nfallback of devices that are not peers) -- beside export + import through a host blob, at the 128-bit and Uint5 key sizes (keys generated on the GPU)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
graft.build(); pkg = graft.load_package()
from oracle_lib import Oracle
import torch
o = Oracle()
for name in ("128", "uint5"):
    p = o.params(name)
    P = pkg.Params(n=p.n, N=p.N, Nbit=p.Nbit, L=p.L, Bgbit=p.Bgbit, basebit=p.basebit, t=p.t)
    ctx = pkg.Context(P)
    rs = np.random.RandomState(1)
    s0 = rs.randint(0, 2, p.n).astype(np.uint32); s1 = rs.randint(0, 2, p.N).astype(np.uint32)
    t0 = time.time(); ctx.keygen_cloud(s0, s1, 2.0e-5 if name == "128" else 7.1e-8, 2.0e-8 if name == "128" else 2.2e-17, seed=5); torch.cuda.synchronize()
    kg = time.time() - t0
    sizes = (ctx.key_size(0), ctx.key_size(1))
    res = {}
    for how in ("d2d", "host-staged"):
        ctx.set_option("clone_force_host", 1 if how == "host-staged" else 0)
        ts = []
        for _ in range(4):
            t0 = time.time(); c2 = ctx.clone_to(0); torch.cuda.synchronize(); ts.append(time.time() - t0)
            path = c2.get_option("clone_path"); c2.close()
        res[how] = (min(ts) * 1e3, path)
    ctx.set_option("clone_force_host", 0)
    t0 = time.time(); blobs = [ctx.key_export(w) for w in (0, 1)]; c3 = pkg.Context(P); [c3.key_import(w, b) for w, b in enumerate(blobs)]; ex = time.time() - t0; c3.close()
    tot = sum(sizes) / 1e6
    print(f"{name}: keys {sizes[0]/1e6:.1f} + {sizes[1]/1e6:.1f} MB; GPU keygen {kg*1e3:.0f} ms; clone same-GPU D2D {res['d2d'][0]:.1f} ms (path {res['d2d'][1]}, incl. context creation and the derived layouts); "
          f"host-staged {res['host-staged'][0]:.1f} ms (path {res['host-staged'][1]}) = {tot/res['host-staged'][0]:.1f} GB/s; export + import through a host blob {ex*1e3:.0f} ms", flush=True)
    ctx.close()
