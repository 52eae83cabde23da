#!/usr/bin/env python3
"""Bit-identity of library variants on ONE box (run on the GPU box):
   python tools/ab_equal.py [--params uint5] [--batches 1,64,300,512,513,1024] libA.so libB.so ...
Each variant runs in its own process (TFHE_HIP_LIB) on the same seeded random key and inputs and prints a SHA-256 of the
blind-rotate accumulators and of the bootstrapped LWE samples per batch size; the script compares them across variants.
Used to accept a kernel refactor that must not change a single output word."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, hashlib, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
pname = sys.argv[1]; batches = [int(x) for x in sys.argv[2].split(",")]
p = pkg.params.BY_NAME[pname]
rs = np.random.RandomState(7)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2*p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n+1)))
res = {}
for B in batches:
    a = torch.from_numpy(rnd((B, p.n+1)).view(np.int32)).cuda()
    lut = torch.from_numpy(rnd((2, p.N)).view(np.int32)).cuda()
    acc = torch.zeros((B, 2, p.N), dtype=torch.int32, device="cuda")
    out = torch.zeros((B, p.n+1), dtype=torch.int32, device="cuda")
    ck.ctx.blind_rotate_batch_dev(a, lut, acc)
    ck.ctx.bootstrap_batch_dev(a, lut, out)
    torch.cuda.synchronize()
    res[str(B)] = [hashlib.sha256(acc.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]]
print(json.dumps(res))
''' % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--params", default="uint5")
ap.add_argument("--batches", default="1,64,300,512,513,1024")
args = ap.parse_args()
got = {}
for l in args.libs:
    env = dict(os.environ)
    if l != "default":
        env["TFHE_HIP_LIB"] = os.path.abspath(l)
    out = subprocess.run([sys.executable, "-c", WORKER, args.params, args.batches], env=env, capture_output=True, text=True, cwd=ROOT)
    try:
        got[l] = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print("FAILED", l, out.stderr[-600:]); sys.exit(1)
ref = got[args.libs[0]]
ok = True
for l in args.libs[1:]:
    for B, h in ref.items():
        same = got[l][B] == h
        ok &= same
        print(f"{os.path.basename(l):24s} vs {os.path.basename(args.libs[0]):24s} params={args.params} B={B:>5s}: {'identical' if same else 'DIFFERENT ' + str(got[l][B]) + ' vs ' + str(h)}")
sys.exit(0 if ok else 1)
