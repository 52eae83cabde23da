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
print("schema fixtures (ORACLE-made, pin nothing) in", out)
