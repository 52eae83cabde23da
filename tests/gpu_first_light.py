"""Quick on-box sanity + timing (not a pytest file): run as `python tests/gpu_first_light.py`."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as g
g.build()
pkg = g.load_package()
from oracle_lib import Oracle
o = Oracle()
p = o.params("128")
rng = o.rng(5)
t = time.time()
s0, s1 = o.keygen_secret(p, rng)
_, bsk = o.keygen_bsk(p, rng, s0, s1, torus=False)
ksk = o.keygen_ksk(p, rng, s0, s1)
print("keygen s", time.time() - t, flush=True)
ck = pkg.CloudKey(pkg.params.Security128Bit, bsk_fourier=bsk, ksk=ksk)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rs = np.random.RandomState(0)
A = rs.randint(0, 2, B); Bb = rs.randint(0, 2, B)
a = o.encrypt_bools(p, rng, A, s0); b = o.encrypt_bools(p, rng, Bb, s0)
walls = []
for it in range(10):
    t = time.time()
    out = ck.ctx.gate_batch("NAND", a, b)
    dt = time.time() - t
    walls.append(dt)
    print(f"iter {it}: {dt*1e3:.1f} ms wall, BR kernel {ck.ctx.last_kernel_ms(0):.2f} ms, KS kernel {ck.ctx.last_kernel_ms(1):.2f} ms, {B/dt:.0f} gates/s", flush=True)
med = sorted(walls)[len(walls) // 2]
print(f"pageable host buffers: median {med*1e3:.2f} ms wall = {B/med:.0f} gates/s")
dec = o.decrypt_bools(p, s0, out)
print("correct:", int((dec == ~(A.astype(bool) & Bb.astype(bool))).sum()), "/", B)
want, _ = o.gate_batch(p, bsk, ksk, "NAND", a[:4], b[:4])
print("bit-exact first 4:", np.array_equal(out[:4], want))
# page-locked operands/outputs (tfhe_host_alloc): the fast path of the host-pointer ABI
pa, pb, po = pkg.PinnedArray(a.shape), pkg.PinnedArray(b.shape), pkg.PinnedArray(a.shape)
pa.array[...] = a; pb.array[...] = b
walls = []
for it in range(10):
    t = time.time()
    ck.ctx.gate_batch("NAND", pa.array, pb.array, out=po.array)
    walls.append(time.time() - t)
med = sorted(walls)[len(walls) // 2]
print(f"page-locked host buffers (tfhe_host_alloc): median {med*1e3:.2f} ms wall = {B/med:.0f} gates/s", flush=True)
print("pinned result identical:", np.array_equal(po.array, out))
