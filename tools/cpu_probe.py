import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
from oracle_lib import Oracle
o = Oracle(); p = o.params("128"); rng = o.rng(1)
s0, s1 = o.keygen_secret(p, rng); _, bf = o.keygen_bsk(p, rng, s0, s1, torus=False); ksk = o.keygen_ksk(p, rng, s0, s1)
a = o.encrypt_bools(p, rng, [1] * 512, s0); b = o.encrypt_bools(p, rng, [0] * 512, s0)
for nt in (1, 8, 16, 32, 64, 128, 256):
    S = max(2 * nt, 4) if nt > 1 else 2
    S = min(S, 512)
    t = time.time(); o.gate_batch(p, bf, ksk, "NAND", a[:S], b[:S], nthreads=nt); dt = time.time() - t
    print(f"threads {nt:3d}: {S} gates in {dt:.2f}s -> {S/dt:.1f} gates/s ({dt/S*nt*1e3:.0f} ms per gate per thread)", flush=True)
