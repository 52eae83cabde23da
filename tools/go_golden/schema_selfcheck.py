#!/usr/bin/env python3
"""Plumbing check for tests/test_go_golden.py WITHOUT Go: writes files with the schema of tools/go_golden/main.go
from the C ORACLE into a scratch directory, so the loaders and shapes of the test module can be exercised:
   python tools/go_golden/schema_selfcheck.py /tmp/gg
   TFHE_GO_GOLDEN_SMALL=/tmp/gg/small TFHE_GO_GOLDEN_BIG=/tmp/gg/big python -m pytest tests/test_go_golden.py -m "not gpu"
These files pin NOTHING (oracle checked against itself) and must never be committed under tests/golden/go/."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
out = sys.argv[1]
small, big = os.path.join(out, "small"), os.path.join(out, "big")
os.makedirs(small, exist_ok=True); os.makedirs(big, exist_ok=True)
o = Oracle(); p = o.params("128")
rng = o.rng(0x7F4E0020); s0, s1 = o.keygen_secret(p, rng)
_, bsk = o.keygen_bsk(p, rng, s0, s1, torus=False, fourier=True); ksk = o.keygen_ksk(p, rng, s0, s1)
rs = np.random.RandomState(1); rnd = lambda s: rs.randint(0, 2**32, size=s, dtype=np.uint64).astype(np.uint32)
np.save(os.path.join(small, "params.npy"), np.array([p.n, p.N, p.Nbit, p.L, p.Bgbit, p.basebit, p.t, o.offset(p)], np.int64))
trl = rnd((2, p.N))
np.save(os.path.join(small, "extprod_trgsw.npy"), bsk[0]); np.save(os.path.join(small, "extprod_in.npy"), trl)
np.save(os.path.join(small, "extprod_out.npy"), o.external_product(p, bsk[0], trl))
K = 4; lwe = rnd(p.n + 1); tv = o.gate_testvec(p); pk = p.small(K); ct = np.concatenate([lwe[:K], lwe[-1:]])
np.save(os.path.join(small, "cmux_trgsw.npy"), bsk[:K]); np.save(os.path.join(small, "cmux_lwe.npy"), lwe)
np.save(os.path.join(small, "cmux_acc.npy"), np.stack([o.blind_rotate(pk, bsk[:K], ct, tv, s) for s in range(K + 1)]))
B = 2; bits = rs.randint(0, 2, B).astype(np.uint8); cts = o.encrypt_bools(p, rng, bits, s0)
np.save(os.path.join(big, "key_lv0.npy"), s0); np.save(os.path.join(big, "key_lv1.npy"), s1)
np.save(os.path.join(big, "bsk_fourier.npy"), bsk); np.save(os.path.join(big, "ksk.npy"), ksk)
np.save(os.path.join(big, "bits.npy"), bits); np.save(os.path.join(big, "lwe_in.npy"), cts)
acc = np.stack([o.blind_rotate(p, bsk, c, tv) for c in cts]); np.save(os.path.join(big, "trlwe_acc.npy"), acc)
np.save(os.path.join(big, "lwe_out.npy"), np.stack([o.key_switch(p, ksk, o.sample_extract(a)) for a in acc]))
gb = rs.randint(0, 2, (3, B)).astype(np.uint8); ga, gbb, gc = (o.encrypt_bools(p, rng, gb[k], s0) for k in range(3))
np.save(os.path.join(big, "gate_bits.npy"), gb)
for nm, v in (("a", ga), ("b", gbb), ("c", gc)): np.save(os.path.join(big, f"gate_{nm}.npy"), v)
for op in ["NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX"]:
    np.save(os.path.join(big, f"gate_{op}.npy"), o.gate_batch(p, bsk, ksk, op, ga, gbb, gc if op == "MUX" else None)[0])
# ---- the programmable-bootstrap seam (Uint5); the key is generated at a REDUCED n (a full Uint5 key-switching key is 1.7 GB):
# tests/test_go_golden.py reads n from uint5_params.npy
p5 = o.params("uint5").small(int(os.environ.get("SELFCHECK_UINT5_N", "24"))); m = 32
big5 = os.path.join(big, "uint5"); os.makedirs(big5, exist_ok=True)
rng5 = o.rng(0x7F4E0021); t0, t1 = o.keygen_secret(p5, rng5)
_, bsk5 = o.keygen_bsk(p5, rng5, t0, t1, torus=False, fourier=True); ksk5 = o.keygen_ksk(p5, rng5, t0, t1)
np.save(os.path.join(small, "uint5_params.npy"), np.array([p5.n, p5.N, p5.Nbit, p5.L, p5.Bgbit, p5.basebit, p5.t, o.offset(p5), m], np.int64))
funcs = [("identity", lambda x: x), ("mod16", lambda x: x % 16), ("ge16", lambda x: int(x >= 16))]
tabs = [o.lut_generate(p5, [f(x) for x in range(m)]) for _, f in funcs]
for (nm, _), t in zip(funcs, tabs): np.save(os.path.join(small, f"uint5_lut_{nm}.npy"), t)
np.save(os.path.join(big5, "key_lv0.npy"), t0); np.save(os.path.join(big5, "key_lv1.npy"), t1)
np.save(os.path.join(big5, "bsk_fourier.npy"), bsk5); np.save(os.path.join(big5, "ksk.npy"), ksk5)
P = 6; msgs = rs.randint(0, m, P).astype(np.int64); which = (np.arange(P) % 3).astype(np.int64)
ins = np.stack([o.encrypt_message(p5, rng5, int(x), m, t0) for x in msgs])
outs = np.stack([o.bootstrap(p5, bsk5, ksk5, ins[i], tabs[which[i]]) for i in range(P)])
np.save(os.path.join(big5, "pbs_msgs.npy"), msgs); np.save(os.path.join(big5, "pbs_lut.npy"), which)
np.save(os.path.join(big5, "pbs_in.npy"), ins); np.save(os.path.join(big5, "pbs_out.npy"), outs)
np.save(os.path.join(big5, "pbs_dec.npy"), np.array([o.decrypt_message(p5, m, t0, r) for r in outs], np.int64))
print("schema fixtures (ORACLE-made, pin nothing) in", out)
