#!/usr/bin/env python3
"""A/B kernel timing across library variants on ONE box (run on the GPU box):
   python tools/ab_bench.py [--rounds 3] [--launches 12] libA.so libB.so ...
Each variant runs in its own process (TFHE_HIP_LIB), variants interleaved per round; prints
mean/min blind-rotate and key-switch kernel time per variant."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
B = int(sys.argv[1]); L = int(sys.argv[2]); pname = sys.argv[3]
p = pkg.params.BY_NAME[pname]
rs = np.random.RandomState(1)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2*p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n+1)))
a = torch.from_numpy(rnd((B, p.n+1)).view(np.int32)).cuda(); b = torch.from_numpy(rnd((B, p.n+1)).view(np.int32)).cuda()
out = torch.empty_like(a)
lut = torch.from_numpy(rnd((2, p.N)).view(np.int32)).cuda()
# gate sets run NAND gates, Uint sets a LUT bootstrap (same two kernels)
step = (lambda: ck.ctx.gate_batch_dev("NAND", a, b, None, out)) if pname in ("80", "110", "128") else (lambda: ck.ctx.bootstrap_batch_dev(a, lut, out))
for _ in range(3): step()
torch.cuda.synchronize()
br, ks = [], []
from go_tfhe_amd import telemetry
smp = telemetry.Sampler(0, period_s=0.005)
with smp:
    for _ in range(L):
        step(); torch.cuda.synchronize()
        br.append(ck.ctx.last_kernel_ms(0)); ks.append(ck.ctx.last_kernel_ms(1))
print(json.dumps({"br": br, "ks": ks, "tel": smp.summary()}))
''' % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--launches", type=int, default=12)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--params", default="128", help="parameter set name (go-tfhe_amd/params.py BY_NAME)")
args = ap.parse_args()
res = {l: {"br": [], "ks": [], "clk": [], "pw": []} for l in args.libs}
for r in range(args.rounds):
    for l in args.libs:
        env = dict(os.environ)
        if l != "default":
            env["TFHE_HIP_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, "-c", WORKER, str(args.batch), str(args.launches), args.params], env=env,
                             capture_output=True, text=True, cwd=ROOT)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print("FAILED", l, out.stderr[-400:]); continue
        res[l]["br"] += d["br"]; res[l]["ks"] += d["ks"]
        if d.get("tel", {}).get("available"):
            res[l]["clk"].append(d["tel"].get("sclk_mhz_mean", 0)); res[l]["pw"].append(d["tel"].get("power_w_max", 0))
for l in args.libs:
    br, ks = res[l]["br"], res[l]["ks"]
    if br:
        tel = ""
        if res[l]["clk"]:
            tel = f" | sclk {sum(res[l]['clk'])/len(res[l]['clk']):.0f} MHz, power max {max(res[l]['pw']):.0f} W"
        print(f"{os.path.basename(l):28s} BR mean {sum(br)/len(br):.3f} min {min(br):.3f} max {max(br):.3f} | KS mean {sum(ks)/len(ks):.3f} min {min(ks):.3f}  (n={len(br)}){tel}")
