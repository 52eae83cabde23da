"""A stand-in for cgo's pseudo-package "C" so that the Go shim (shim/go/**) can be EXECUTED by gointerp.py where there is neither a Go
toolchain nor a GPU: every C.tfhe_* entry point the shim calls is implemented here on top of the CPU oracle (tests/oracle_lib.py), with
the semantics include/tfhe_hip.h documents (0 / negative return codes, tfhe_last_error, caller-owned outputs written through the
pointers the shim passes).  Test infrastructure: what it validates is the SHIM's own logic -- flattening []*tlwe.TLWELv0 into the ABI's
[B][n+1] planes and back, op codes, the key registry, CloudKeySet's contiguous shards and goroutine fan-out -- by comparing what the
shim returns with what the reference's own gates return (both under the interpreter).  It says nothing about the GPU library."""
import os
import re

import numpy as np

import gointerp as gi

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OPNAMES = ["NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX"]


class OracleBackend:
    """The entry points' arithmetic on the CPU oracle (CPU tier: validates the shim's logic, not the GPU library)."""

    def __init__(self, oracle, device_count=2):
        self.o, self.ndev = oracle, device_count

    def device_count(self):
        return self.ndev

    def params(self, f):
        n, N, L, Bg, bb, t = (int(f[k]) for k in ("n", "N", "L", "Bgbit", "basebit", "t"))
        for name in ("80", "110", "128", "uint5", "uint1", "uint3", "uint4", "uint2"):
            p = self.o.params(name)
            if (p.N, p.L, p.Bgbit, p.basebit, p.t) == (N, L, Bg, bb, t):
                return p.small(n)
        raise MockError(-1, f"unsupported parameter shape N={N} L={L} Bgbit={Bg}")

    def create(self, p, device):
        return {"bsk": None, "ksk": None}

    def destroy(self, h):
        pass

    def clone(self, h, device):
        return {"bsk": h["bsk"], "ksk": h["ksk"]}, (1 if True else 2)

    def load_bsk(self, h, p, arr):
        h["bsk"] = arr

    def load_ksk(self, h, p, arr):
        h["ksk"] = arr

    def _keys(self, h, need_ksk=True):
        if h["bsk"] is None or (need_ksk and h["ksk"] is None):
            raise MockError(-2, "cloud key not loaded")
        return h["bsk"], h["ksk"]

    def gate_batch(self, h, p, codes, A, B, C):
        bsk, ksk = self._keys(h)
        return self.o.gate_batch(p, bsk, ksk, codes, A, B, C)[0]

    def bootstrap_batch(self, h, p, X, T):
        bsk, ksk = self._keys(h)
        return self.o.bootstrap_batch(p, bsk, ksk, X, T)[0]

    def blind_rotate_batch(self, h, p, X, T, nsteps):
        bsk, _ = self._keys(h, need_ksk=False)
        return np.stack([self.o.blind_rotate(p, bsk, X[i], T, int(nsteps)) for i in range(len(X))])

    def gate_testvec(self, p):
        return self.o.gate_testvec(p)

    def offset(self, h, p):
        return self.o.offset(p)

    def external_product_with(self, h, p, gsw, off, X):
        return np.stack([self.o.external_product_at_offset(p, gsw, X[i], off) for i in range(len(X))])

    def cmux_with(self, h, p, gsw, off, X0, X1):
        return np.stack([self.o.cmux_at_offset(p, gsw, X0[i], X1[i], off) for i in range(len(X0))])

    def sample_extract(self, h, p, X, k):
        return np.stack([self.o.sample_extract(np.ascontiguousarray(X[i]), int(k)) for i in range(len(X))])

    def keyswitch(self, h, p, X):
        if h["ksk"] is None:
            raise MockError(-2, "key-switching key not loaded")
        return np.stack([self.o.key_switch(p, h["ksk"], np.ascontiguousarray(X[i])) for i in range(len(X))])


class LibBackend:
    """The REAL library: every entry point goes to libtfhe_hip.so through the Python binding's ctypes layer (go-tfhe_amd/_binding.py), i.e.
    the Go shim, executed by the interpreter, drives the GPU (-m gpu tier)."""

    def __init__(self, pkg, oracle):
        self.pkg, self.o = pkg, oracle

    def device_count(self):
        import ctypes
        n = ctypes.c_int()
        assert self.pkg.load_library().tfhe_device_count(ctypes.byref(n)) == 0
        return n.value

    def params(self, f):
        return self.pkg.Params(**{k: int(f[k]) for k in ("n", "N", "Nbit", "L", "Bgbit", "basebit", "t")})

    def _wrap(self, fn):
        try:
            return fn()
        except self.pkg.TfheError as e:
            raise MockError(getattr(e, "code", -1), str(e))

    def create(self, p, device):
        return self._wrap(lambda: self.pkg.Context(p, device))

    def destroy(self, h):
        h.close()

    def clone(self, h, device):
        c = self._wrap(lambda: h.clone_to(device))
        return c, c.get_option("clone_path")

    def load_bsk(self, h, p, arr):
        self._wrap(lambda: h.load_bsk_fourier(arr))

    def load_ksk(self, h, p, arr):
        self._wrap(lambda: h.load_ksk(arr))

    def gate_batch(self, h, p, codes, A, B, C):
        return self._wrap(lambda: h.gate_batch(np.ascontiguousarray(codes, np.uint8), A, B, C))

    def bootstrap_batch(self, h, p, X, T):
        return self._wrap(lambda: h.bootstrap_batch(X, T))

    def blind_rotate_batch(self, h, p, X, T, nsteps):
        return self._wrap(lambda: h.blind_rotate_batch(X, T, int(nsteps)))

    def gate_testvec(self, p):
        return None                                         # NULL = the library's own gate test vector

    def keygen(self, h, p, s0, s1, a0, a1):
        self._wrap(lambda: h.keygen_cloud(s0, s1, a0, a1, None))

    def offset(self, h, p):
        return self._wrap(lambda: h.decomposition_offset())

    def external_product_with(self, h, p, gsw, off, X):
        return self._wrap(lambda: h.external_product_with(gsw, X, off))

    def cmux_with(self, h, p, gsw, off, X0, X1):
        return self._wrap(lambda: h.cmux_with(gsw, X0, X1, off))

    def sample_extract(self, h, p, X, k):
        return self._wrap(lambda: h.sample_extract_batch(X, int(k)))

    def keyswitch(self, h, p, X):
        return self._wrap(lambda: h.keyswitch_batch(X))

    def key_size(self, h, which):
        return self._wrap(lambda: h.key_size(which))

    def key_export(self, h, which):
        return self._wrap(lambda: h.key_export(which))

    def key_import(self, h, which, blob):
        self._wrap(lambda: h.key_import(which, blob))


class MockC:
    def __init__(self, interp, oracle=None, device_count=2, backend=None):
        self.I = interp
        self.be = backend or OracleBackend(oracle, device_count)
        self.o, self.ndev = oracle, self.be.device_count()
        self.err = ""
        self.ctxs = []
        self.calls = []                      # (name, detail) log: the tests look at it (which device ran how many items)
        pkg = gi.Pkg("C", "C")
        pkg.initialised = True
        interp.pkgs["C"] = pkg
        T = gi.BASIC_RT
        for n in ("int", "int32_t", "size_t", "uint64_t", "int64_t", "long"):
            pkg.native[n] = T["int"]
        pkg.native["uint32_t"], pkg.native["uint8_t"], pkg.native["double"], pkg.native["char"] = T["uint32"], T["uint8"], T["float64"], T["int8"]
        self.PARAMS = gi.RT("struct", fields=[(f, T["int"]) for f in ("n", "N", "Nbit", "L", "Bgbit", "basebit", "t")])
        self.PARAMS._stub = "tfhe_params"
        self.CTX = gi.RT("struct", fields=[("id", T["int"])])
        self.CTX._stub = "tfhe_ctx"
        pkg.native["tfhe_params"], pkg.native["tfhe_ctx"] = self.PARAMS, self.CTX
        hdr = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
        hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
        for m in re.finditer(r"enum\s*\{(.*?)\}", hdr, flags=re.S):
            nxt = 0
            for em in re.finditer(r"(\w+)\s*(?:=\s*(-?\d+))?\s*(?:,|$)", m.group(1).strip()):
                nxt = int(em.group(2)) if em.group(2) is not None else nxt
                pkg.native[em.group(1)] = nxt
                nxt += 1
        for m in re.finditer(r"#define\s+(TFHE_\w+)\s+(-?\d+)", hdr):
            pkg.native[m.group(1)] = int(m.group(2))
        for name in [n for n in dir(self) if n.startswith("tfhe_")]:
            pkg.native[name] = gi.Builtin(self._wrap(getattr(self, name), name), name)
        pkg.native["GoString"] = gi.Builtin(lambda a: a[0], "GoString")

    def _wrap(self, fn, name):
        def call(a):
            try:
                return fn(*a)
            except MockError as e:
                self.err = str(e)
                return e.code
        return call

    # ---- helpers
    def ctx(self, h):
        if h is None:
            raise MockError(-1, "null context")
        c = self.ctxs[int(h.v.f["id"])]
        if c is None:
            raise MockError(-1, "context already destroyed")
        return c

    @staticmethod
    def read(ptr, count, dtype):
        if ptr is None:
            return None
        if ptr.i + count > len(ptr.a):
            raise MockError(-1, f"the shim handed C {len(ptr.a) - ptr.i} elements where the ABI reads {count}")
        return np.array(ptr.a[ptr.i:ptr.i + count], dtype=dtype)

    @staticmethod
    def write(ptr, arr):
        flat = np.ascontiguousarray(arr, np.uint32).reshape(-1)
        if ptr.i + flat.size > len(ptr.a):
            raise MockError(-1, "output buffer shorter than the ABI writes")
        ptr.a[ptr.i:ptr.i + flat.size] = [np.uint32(x) for x in flat.tolist()]

    # ---- the entry points the shim uses (include/tfhe_hip.h)
    def tfhe_last_error(self):
        return self.err

    def tfhe_device_count(self, out):
        gi.ptr_store(out, self.ndev)
        return 0

    def _new(self, rec, out):
        self.ctxs.append(rec)
        gi.ptr_store(out, gi.GoPtr(gi.GoStruct(self.CTX, {"id": len(self.ctxs) - 1})))

    def tfhe_ctx_create(self, pptr, device, out):
        p = self.be.params(gi.ptr_load(pptr).f)
        if isinstance(self.be, OracleBackend) and not 0 <= int(device) < self.ndev:
            raise MockError(-1, f"device {device} not present ({self.ndev} visible)")
        self._new({"p": p, "device": int(device), "h": self.be.create(p, int(device)), "clone_path": 0}, out)
        self.calls.append(("ctx_create", int(device)))
        return 0

    def tfhe_ctx_destroy(self, h):
        if h is not None:
            self.be.destroy(self.ctx(h)["h"])
            self.ctxs[int(h.v.f["id"])] = None
        return 0

    def tfhe_ctx_clone_to(self, h, device, out):
        src = self.ctx(h)
        if isinstance(self.be, OracleBackend) and not 0 <= int(device) < self.ndev:
            raise MockError(-1, f"device {device} not present ({self.ndev} visible)")
        hh, path = self.be.clone(src["h"], int(device))
        if isinstance(self.be, OracleBackend):
            path = 1 if int(device) == src["device"] else 2
        self._new({"p": src["p"], "device": int(device), "h": hh, "clone_path": path}, out)
        self.calls.append(("clone_to", int(device)))
        return 0

    def tfhe_ctx_get_option(self, h, opt, out):
        c = self.ctx(h)
        if int(opt) != self.I.pkgs["C"].native["TFHE_OPT_CLONE_PATH"]:
            raise MockError(-1, f"option {opt} is not mocked")
        gi.ptr_store(out, c["clone_path"])
        return 0

    def tfhe_load_bsk_fourier(self, h, ptr):
        c = self.ctx(h)
        p = c["p"]
        self.be.load_bsk(c["h"], p, self.read(ptr, p.n * 2 * p.L * 2 * p.N, np.float64).reshape(p.n, 2 * p.L, 2, p.N))
        self.calls.append(("load_bsk", c["device"]))
        return 0

    def tfhe_load_ksk(self, h, ptr):
        c = self.ctx(h)
        p = c["p"]
        rows = p.N * p.t * (1 << p.basebit)
        self.be.load_ksk(c["h"], p, self.read(ptr, rows * (p.n + 1), np.uint32).reshape(rows, p.n + 1))
        self.calls.append(("load_ksk", c["device"]))
        return 0

    def tfhe_keygen_cloud_seeded(self, h, s0, s1, a0, a1, seed):
        c = self.ctx(h)
        p = c["p"]
        if seed is not None:
            raise MockError(-1, "the shim passes a nil seed (OS entropy)")
        self.be.keygen(c["h"], p, self.read(s0, p.n, np.uint32), self.read(s1, p.N, np.uint32), float(a0), float(a1))
        self.calls.append(("keygen", c["device"]))
        return 0

    def tfhe_key_size(self, h, which, out):
        gi.ptr_store(out, int(self.be.key_size(self.ctx(h)["h"], int(which))))
        return 0

    def tfhe_key_export(self, h, which, dst):
        blob = self.be.key_export(self.ctx(h)["h"], int(which))
        if dst.i + blob.size > len(dst.a):
            raise MockError(-1, "export buffer shorter than tfhe_key_size")
        dst.a[dst.i:dst.i + blob.size] = [np.uint8(x) for x in blob.tolist()]
        self.calls.append(("key_export", int(which), int(blob.size)))
        return 0

    def tfhe_key_import(self, h, which, src, nbytes):
        self.be.key_import(self.ctx(h)["h"], int(which), self.read(src, int(nbytes), np.uint8))
        self.calls.append(("key_import", int(which), int(nbytes)))
        return 0

    def tfhe_gate_batch(self, h, ops, op_uniform, a, b, cc, out, B):
        c = self.ctx(h)
        p = c["p"]
        B, n1 = int(B), p.n + 1
        A, Bb = self.read(a, B * n1, np.uint32).reshape(B, n1), self.read(b, B * n1, np.uint32).reshape(B, n1)
        Cc = self.read(cc, B * n1, np.uint32).reshape(B, n1) if cc is not None else None
        if ops is not None:
            codes = self.read(ops, B, np.uint8)
        else:
            if not 0 <= int(op_uniform) <= 10:
                raise MockError(-1, f"bad op code {op_uniform}")
            codes = np.full(B, int(op_uniform), np.uint8)
        if (codes == 10).any() and Cc is None:
            raise MockError(-1, "MUX needs the third operand")
        self.write(out, self.be.gate_batch(c["h"], p, codes, A, Bb, Cc))
        self.calls.append(("gate_batch", c["device"], B))
        return 0

    def tfhe_bootstrap_batch(self, h, inp, tv, per_item, out, B):
        c = self.ctx(h)
        p = c["p"]
        B, n1 = int(B), p.n + 1
        X = self.read(inp, B * n1, np.uint32).reshape(B, n1)
        if tv is None:
            T = self.be.gate_testvec(p)
        elif int(per_item):
            T = self.read(tv, B * 2 * p.N, np.uint32).reshape(B, 2, p.N)
        else:
            T = self.read(tv, 2 * p.N, np.uint32).reshape(2, p.N)
        self.write(out, self.be.bootstrap_batch(c["h"], p, X, T))
        self.calls.append(("bootstrap_batch", c["device"], B))
        return 0

    def tfhe_blind_rotate_batch(self, h, inp, tv, per_item, out, B, nsteps):
        c = self.ctx(h)
        p = c["p"]
        B, n1 = int(B), p.n + 1
        X = self.read(inp, B * n1, np.uint32).reshape(B, n1)
        T = self.be.gate_testvec(p) if tv is None else self.read(tv, 2 * p.N, np.uint32).reshape(2, p.N)
        self.write(out, self.be.blind_rotate_batch(c["h"], p, X, T, int(nsteps)))
        self.calls.append(("blind_rotate_batch", c["device"], B))
        return 0


    # ---- the trgsw / trlwe seams with caller-supplied operands
    def tfhe_ctx_decomposition_offset(self, h, out):
        c = self.ctx(h)
        gi.ptr_store(out, np.uint32(self.be.offset(c["h"], c["p"])))
        return 0

    def _gsw(self, p, ptr):
        if ptr is None:
            raise MockError(-1, "null TRGSW operand")
        return self.read(ptr, 2 * p.L * 2 * p.N, np.float64).reshape(2 * p.L, 2, p.N)

    def tfhe_external_product_with(self, h, gsw, off, inp, out, B):
        c = self.ctx(h)
        p, B = c["p"], int(B)
        X = self.read(inp, B * 2 * p.N, np.uint32).reshape(B, 2, p.N)
        self.write(out, self.be.external_product_with(c["h"], p, self._gsw(p, gsw), int(off), X))
        self.calls.append(("external_product_with", c["device"], B))
        return 0

    def tfhe_cmux_with(self, h, gsw, off, ct0, ct1, out, B):
        c = self.ctx(h)
        p, B = c["p"], int(B)
        X0 = self.read(ct0, B * 2 * p.N, np.uint32).reshape(B, 2, p.N)
        X1 = self.read(ct1, B * 2 * p.N, np.uint32).reshape(B, 2, p.N)
        self.write(out, self.be.cmux_with(c["h"], p, self._gsw(p, gsw), int(off), X0, X1))
        self.calls.append(("cmux_with", c["device"], B))
        return 0

    def tfhe_sample_extract_batch(self, h, inp, k, out, B):
        c = self.ctx(h)
        p, B = c["p"], int(B)
        if not 0 <= int(k) < p.N:
            raise MockError(-1, f"sample-extract index {k} outside [0, {p.N})")
        X = self.read(inp, B * 2 * p.N, np.uint32).reshape(B, 2, p.N)
        self.write(out, self.be.sample_extract(c["h"], p, X, int(k)))
        self.calls.append(("sample_extract_batch", c["device"], B))
        return 0

    def tfhe_keyswitch_batch(self, h, inp, out, B):
        c = self.ctx(h)
        p, B = c["p"], int(B)
        X = self.read(inp, B * (p.N + 1), np.uint32).reshape(B, p.N + 1)
        self.write(out, self.be.keyswitch(c["h"], p, X))
        self.calls.append(("keyswitch_batch", c["device"], B))
        return 0


class MockError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code
