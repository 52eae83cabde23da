"""Run by tests/test_gpu_abi_misuse.py in a subprocess (a crash must not take the test session down): calls every entry point of
include/tfhe_hip.h straight through ctypes -- none of the Python wrapper's argument checks in between -- with (1) a NULL context, (2) a live
context (keys loaded) and NULL for every other pointer, (3) a live context, valid buffers and a negative batch size, and prints what came back.
The C ABI promises: never throws, never crashes on a NULL or a negative count, returns a negative code and leaves a message in tfhe_last_error."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

graft.build()
pkg = graft.load_package()
from oracle_lib import Oracle  # noqa: E402

o = Oracle()
lib = pkg.load_library()
p = o.params("128").small(8)
rng = o.rng(5)
s0, s1 = o.keygen_secret(p, rng)
_, bsk = o.keygen_bsk(p, rng, s0, s1, torus=False, fourier=True)
ksk = o.keygen_ksk(p, rng, s0, s1)
ck = pkg.CloudKey(pkg.Params(n=p.n, N=p.N, Nbit=p.Nbit, L=p.L, Bgbit=p.Bgbit, basebit=p.basebit, t=p.t), bsk_fourier=bsk, ksk=ksk)
h = ck.ctx._h

NO_CTX = {"tfhe_device_count", "tfhe_ctx_create", "tfhe_host_alloc", "tfhe_host_free", "tfhe_last_error", "tfhe_build_flavor"}
MAY_ACCEPT_NULL = {"tfhe_ctx_destroy", "tfhe_host_free"}          # destroying / freeing nothing is not an error
results = {}


def zero(t):
    if t in (C.c_int, C.c_uint32, C.c_uint64, C.c_size_t):
        return t(0)
    if t is C.c_double:
        return t(0.0)
    return None                                        # every pointer kind: NULL


for name in pkg.declared_symbols():
    fn = getattr(lib, name)
    if name in ("tfhe_last_error", "tfhe_build_flavor", "tfhe_ctx_destroy"):          # no status code to sweep
        continue
    at = fn.argtypes
    row = {}
    if name not in NO_CTX:
        row["null_ctx"] = int(fn(*[zero(t) for t in at]))
        row["null_ctx_msg"] = bool(lib.tfhe_last_error())
        # a live context, every other pointer NULL, every count 1
        args = [h] + [(t(1) if t in (C.c_int, C.c_size_t) else zero(t)) for t in at[1:]]
        if name in ("tfhe_ctx_sync", "tfhe_timing_enable", "tfhe_ctx_reserve", "tfhe_ctx_reserve_extended", "tfhe_ctx_set_option"):
            row["null_args"] = "n/a (no pointer besides the context)"
        else:
            row["null_args"] = int(fn(*args))
        # ... and a negative count where the function takes one
        if any(t is C.c_int for t in at[1:]) and name.endswith(("_batch", "_batch_dev")):
            buf = np.zeros(4 * 2 * p.N + 64, np.uint32)
            ptr = buf.ctypes.data_as(C.POINTER(C.c_uint32))
            neg = [h]
            for t in at[1:]:
                if t is C.c_int:
                    neg.append(C.c_int(-1))
                elif t is C.c_void_p:
                    neg.append(None if name.endswith("_dev") else C.c_void_p(buf.ctypes.data))
                elif t in (C.POINTER(C.c_uint32),):
                    neg.append(ptr)
                elif t is C.POINTER(C.c_double):
                    neg.append(buf.ctypes.data_as(C.POINTER(C.c_double)))
                elif t is C.POINTER(C.c_uint8):
                    neg.append(buf.ctypes.data_as(C.POINTER(C.c_uint8)))
                else:
                    neg.append(zero(t))
            row["negative_counts"] = int(fn(*neg))
    else:
        row["null_args"] = int(fn(*[zero(t) for t in at])) if name != "tfhe_host_free" else int(fn(None))
    results[name] = row
# destroy(NULL) and the context still works afterwards
results["tfhe_ctx_destroy"] = {"null_ctx": int(lib.tfhe_ctx_destroy(None))}
a = o.encrypt_bools(p, rng, np.array([1, 0], np.uint8), s0)
out = ck.ctx.gate_batch("NAND", a, a[::-1].copy())
results["_context_still_works"] = bool(np.array_equal(o.decrypt_bools(p, s0, out), [True, True]))
ck.close()
print("SWEEP " + json.dumps(results))
