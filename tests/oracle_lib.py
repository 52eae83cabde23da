"""ctypes view of oracle/libtfhe_oracle.so -- the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(ORACLE_DIR, "libtfhe_oracle.so")

PARAM_SETS = {"80": 0, "110": 1, "128": 2, "uint5": 3, "uint6": 3, "uint1": 4, "uint3": 5, "uint4": 6, "uint7": 7, "uint8": 7, "uint2": 8}
OPS = {"NAND": 0, "AND": 1, "OR": 2, "XOR": 3, "XNOR": 4, "NOR": 5,
       "ANDNY": 6, "ANDYN": 7, "ORNY": 8, "ORYN": 9, "MUX": 10}


class Params(C.Structure):
    _fields_ = [("n", C.c_int32), ("N", C.c_int32), ("Nbit", C.c_int32), ("L", C.c_int32),
                ("Bgbit", C.c_int32), ("basebit", C.c_int32), ("t", C.c_int32),
                ("alpha_lv0", C.c_double), ("alpha_lv1", C.c_double)]

    @property
    def base(self):
        return 1 << self.basebit

    @property
    def ksk_rows(self):
        return self.N * self.t * self.base

    def small(self, n):
        """Same ring/gadget, shorter LWE dimension (keeps tests fast)."""
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        q.n = n
        return q


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("have_spare", C.c_int), ("spare", C.c_double)]


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _u32p(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f64p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    def __init__(self):
        if not os.path.exists(_LIB):
            build()
        self.lib = lib = C.CDLL(_LIB)
        lib.orc_f64_to_torus.restype = C.c_uint32
        lib.orc_f64_to_torus.argtypes = [C.c_double]
        lib.orc_decomposition_offset.restype = C.c_uint32
        lib.orc_fft_new.restype = C.c_void_p
        lib.orc_fft_new.argtypes = [C.c_int]
        lib.orc_fft_free.argtypes = [C.c_void_p]
        lib.orc_tlwe_phase.restype = C.c_uint32
        lib.orc_rng_u64.restype = C.c_uint64
        self._fft = {}

    # -- params / scalars
    def params(self, name):
        p = Params()
        assert self.lib.orc_get_params(PARAM_SETS[name], C.byref(p)) == 0
        return p

    def f64_to_torus(self, d):
        return self.lib.orc_f64_to_torus(d)

    def offset(self, p):
        return self.lib.orc_decomposition_offset(C.byref(p))

    def fft(self, N):
        if N not in self._fft:
            self._fft[N] = C.c_void_p(self.lib.orc_fft_new(N))
        return self._fft[N]

    def fft_twiddles(self, N):
        """(tw, twInv) of the evaluator for ring degree N as complex128 arrays of N/2 - 1 entries (poly_evaluator.go:114-143)."""
        M = N // 2
        arrs = [np.empty(M - 1, np.float64) for _ in range(4)]
        self.lib.orc_fft_twiddles(self.fft(N), *[_f64p(a) for a in arrs])
        return arrs[0] + 1j * arrs[1], arrs[2] + 1j * arrs[3]

    def rng(self, seed):
        r = Rng()
        self.lib.orc_rng_seed(C.byref(r), C.c_uint64(seed))
        return r

    # -- poly layer
    def to_fourier(self, poly):
        N = poly.shape[-1]
        out = np.empty(N, np.float64)
        self.lib.orc_to_fourier(self.fft(N), _u32p(poly), _f64p(out))
        return out

    def to_poly(self, fp, want_pre=False):
        N = fp.shape[-1]
        fp = fp.copy()
        out = np.empty(N, np.uint32)
        pre = np.empty(N, np.float64) if want_pre else None
        self.lib.orc_to_poly(self.fft(N), _f64p(fp), _u32p(out), _f64p(pre) if want_pre else None)
        return (out, pre) if want_pre else out

    def decompose(self, p, poly):
        out = np.empty((p.L, p.N), np.uint32)
        self.lib.orc_decompose(C.byref(p), _u32p(poly), C.c_uint32(self.offset(p)), _u32p(out))
        return out

    def poly_mul_xk(self, a, k):
        out = np.empty_like(a)
        self.lib.orc_poly_mul_xk(a.shape[0], _u32p(a), C.c_int(k), _u32p(out))
        return out

    def negacyclic_exact(self, a, b):
        out = np.empty_like(b)
        self.lib.orc_negacyclic_exact(a.shape[0], _u32p(a), _u32p(b), _u32p(out))
        return out

    # -- ciphertext path
    def external_product(self, p, bsk_i, ct):
        out = np.empty((2, p.N), np.uint32)
        self.lib.orc_external_product(C.byref(p), self.fft(p.N), _f64p(bsk_i), _u32p(ct), _u32p(out))
        return out

    def external_product_at_offset(self, p, gsw, ct, offset):
        """ExternalProductAssign (evaluator.go:50-81) composed from the restated primitives with an ARBITRARY decomposition offset (the
        reference takes it as an argument; orc_external_product uses the cloud key's).  gsw: [2L][2][N] float64."""
        N, L = p.N, p.L
        rows = np.ascontiguousarray(gsw, np.float64).reshape(2 * L, 2, N)
        acc = [np.zeros(N, np.float64), np.zeros(N, np.float64)]
        for part in range(2):
            digits = np.empty((L, N), np.uint32)
            self.lib.orc_decompose(C.byref(p), _u32p(np.ascontiguousarray(ct[part])), C.c_uint32(int(offset)), _u32p(digits))
            for l in range(L):
                f = self.to_fourier(np.ascontiguousarray(digits[l]))
                for ab in range(2):
                    self.lib.orc_fourier_mul_add(N, _f64p(f), _f64p(np.ascontiguousarray(rows[part * L + l, ab])), _f64p(acc[ab]))
        return np.stack([self.to_poly(acc[0]), self.to_poly(acc[1])])

    def cmux_at_offset(self, p, gsw, ct0, ct1, offset):
        """CMuxAssign (evaluator.go:85-106) with an arbitrary decomposition offset: ct0 + gsw (x) (ct1 - ct0)."""
        d = (np.asarray(ct1, np.uint32) - np.asarray(ct0, np.uint32)).astype(np.uint32)
        return (np.asarray(ct0, np.uint32) + self.external_product_at_offset(p, gsw, d, offset)).astype(np.uint32)

    def external_product_exact(self, p, bsk_i_torus, ct):
        out = np.empty((2, p.N), np.uint32)
        self.lib.orc_external_product_exact(C.byref(p), _u32p(bsk_i_torus), _u32p(ct), _u32p(out))
        return out

    def cmux(self, p, bsk_i, ct0, ct1):
        out = np.empty((2, p.N), np.uint32)
        self.lib.orc_cmux(C.byref(p), self.fft(p.N), _f64p(bsk_i), _u32p(ct0), _u32p(ct1), _u32p(out))
        return out

    def blind_rotate(self, p, bsk, ct, tv, nsteps=-1):
        out = np.empty((2, p.N), np.uint32)
        self.lib.orc_blind_rotate(C.byref(p), self.fft(p.N), _f64p(bsk), _u32p(ct), _u32p(tv),
                                  C.c_int(nsteps), _u32p(out))
        return out

    def blind_rotate_exact(self, p, bsk_torus, ct, tv, nsteps=-1):
        out = np.empty((2, p.N), np.uint32)
        self.lib.orc_blind_rotate_exact(C.byref(p), _u32p(bsk_torus), _u32p(ct), _u32p(tv),
                                        C.c_int(nsteps), _u32p(out))
        return out

    def blind_rotate_extended(self, p, bsk, ct, lut, nsteps=-1):
        """Blind rotation through an extended lookup table lut [ext][2][N] (LookUpTableSize = ext*N), composed from the
        restated primitives: acc_k <- CMux(bsk[i], acc_k, X^(q + [k<r]) acc_((k-r) mod ext)) with a = ext*q + r the
        mod-switch of ct[i] to [0, 2 ext N).  ext = 1 is BlindRotateAssign (evaluator.go:110-135).  Returns [ext][2][N]."""
        ext, N = lut.shape[0], p.N
        big2 = 2 * ext * N
        ms = lambda x: ((int(x) * big2 + (1 << 31)) >> 32) % big2

        def rotate(acc, a):
            q, r = divmod(a, ext)
            out = np.empty_like(acc)
            for k in range(ext):
                src, s = (k - r) % ext, q + (1 if k < r else 0)
                for part in range(2):
                    out[k, part] = self.poly_mul_xk(np.ascontiguousarray(acc[src, part]), s)
            return out

        acc = rotate(np.ascontiguousarray(lut, np.uint32), (big2 - ms(ct[p.n])) % big2)
        steps = p.n if nsteps < 0 else nsteps
        for i in range(steps):
            rot = rotate(acc, ms(ct[i]))
            acc = np.stack([self.cmux(p, bsk[i], np.ascontiguousarray(acc[k]), np.ascontiguousarray(rot[k])) for k in range(ext)])
        return acc

    def bootstrap_extended(self, p, bsk, ksk, ct, lut):
        acc = self.blind_rotate_extended(p, bsk, ct, lut)
        return self.key_switch(p, ksk, self.sample_extract(np.ascontiguousarray(acc[0])))

    def sample_extract(self, trlwe, k=0):
        N = trlwe.shape[-1]
        out = np.empty(N + 1, np.uint32)
        self.lib.orc_sample_extract(N, _u32p(trlwe), C.c_int(k), _u32p(out))
        return out

    def key_switch(self, p, ksk, lv1):
        out = np.empty(p.n + 1, np.uint32)
        self.lib.orc_key_switch(C.byref(p), _u32p(ksk), _u32p(lv1), _u32p(out))
        return out

    def bootstrap(self, p, bsk, ksk, ct, tv):
        out = np.empty(p.n + 1, np.uint32)
        self.lib.orc_bootstrap(C.byref(p), self.fft(p.N), _f64p(bsk), _u32p(ksk), _u32p(ct), _u32p(tv), _u32p(out))
        return out

    def bootstrap_batch(self, p, bsk, ksk, cts, tv, nthreads=0):
        B = cts.shape[0]
        per_item = 1 if tv.ndim == 3 else 0
        out = np.empty((B, p.n + 1), np.uint32)
        used = self.lib.orc_bootstrap_batch(C.byref(p), _f64p(bsk), _u32p(ksk), _u32p(cts), _u32p(tv),
                                            C.c_int(per_item), _u32p(out), C.c_int(B), C.c_int(nthreads))
        return out, used

    # -- gates
    def gate_prepare(self, p, op, a, b):
        out = np.empty(p.n + 1, np.uint32)
        assert self.lib.orc_gate_prepare(C.byref(p), OPS[op], _u32p(a), _u32p(b), _u32p(out)) == 0
        return out

    def gate_testvec(self, p):
        tv = np.empty((2, p.N), np.uint32)
        self.lib.orc_gate_testvec(C.byref(p), _u32p(tv))
        return tv

    def gate(self, p, bsk, ksk, op, a, b, c=None):
        out = np.empty(p.n + 1, np.uint32)
        rc = self.lib.orc_gate(C.byref(p), self.fft(p.N), _f64p(bsk), _u32p(ksk), OPS[op], _u32p(a), _u32p(b),
                               _u32p(c) if c is not None else None, _u32p(out))
        assert rc == 0
        return out

    def gate_batch(self, p, bsk, ksk, ops, a, b, c=None, nthreads=0):
        B = a.shape[0]
        out = np.empty((B, p.n + 1), np.uint32)
        if isinstance(ops, str):
            uni, opp = OPS[ops], None
        else:
            ops = np.ascontiguousarray(ops, np.uint8)
            uni, opp = -1, ops.ctypes.data_as(C.POINTER(C.c_uint8))
        used = self.lib.orc_gate_batch(C.byref(p), _f64p(bsk), _u32p(ksk), opp, C.c_int(uni), _u32p(a), _u32p(b),
                                       _u32p(c) if c is not None else None, _u32p(out), C.c_int(B), C.c_int(nthreads))
        return out, used

    # -- harness
    def keygen_secret(self, p, rng):
        s0 = np.empty(p.n, np.uint32)
        s1 = np.empty(p.N, np.uint32)
        self.lib.orc_keygen_secret(C.byref(p), C.byref(rng), _u32p(s0), _u32p(s1))
        return s0, s1

    def encrypt_bool(self, p, rng, bit, s0):
        ct = np.empty(p.n + 1, np.uint32)
        self.lib.orc_tlwe_encrypt_bool(C.byref(p), C.byref(rng), C.c_int(int(bit)), _u32p(s0), _u32p(ct))
        return ct

    def encrypt_bools(self, p, rng, bits, s0):
        return np.stack([self.encrypt_bool(p, rng, b, s0) for b in bits])

    def decrypt_bool(self, p, s0, ct):
        return bool(self.lib.orc_tlwe_decrypt_bool(C.byref(p), _u32p(s0), _u32p(ct)))

    def decrypt_bools(self, p, s0, cts):
        return np.array([self.decrypt_bool(p, s0, np.ascontiguousarray(c)) for c in cts])

    def phase(self, p, s0, ct):
        return self.lib.orc_tlwe_phase(C.byref(p), _u32p(s0), _u32p(ct))

    def encrypt_message(self, p, rng, msg, modulus, s0):
        ct = np.empty(p.n + 1, np.uint32)
        self.lib.orc_tlwe_encrypt_message(C.byref(p), C.byref(rng), C.c_int(msg), C.c_int(modulus), _u32p(s0), _u32p(ct))
        return ct

    def decrypt_message(self, p, modulus, s0, ct):
        return self.lib.orc_tlwe_decrypt_message(C.byref(p), C.c_int(modulus), _u32p(s0), _u32p(ct))

    def keygen_bsk(self, p, rng, s0, s1, torus=True, fourier=True):
        shape = (p.n, 2 * p.L, 2, p.N)
        bt = np.empty(shape, np.uint32) if torus else None
        bf = np.empty(shape, np.float64) if fourier else None
        self.lib.orc_keygen_bsk(C.byref(p), C.byref(rng), _u32p(s0), _u32p(s1),
                                _u32p(bt) if torus else None, _f64p(bf) if fourier else None)
        return bt, bf

    def keygen_ksk(self, p, rng, s0, s1):
        ksk = np.empty((p.ksk_rows, p.n + 1), np.uint32)
        self.lib.orc_keygen_ksk(C.byref(p), C.byref(rng), _u32p(s0), _u32p(s1), _u32p(ksk))
        return ksk

    def lut_generate(self, p, table):
        table = np.ascontiguousarray(table, np.int32)
        tv = np.empty((2, p.N), np.uint32)
        self.lib.orc_lut_generate(C.byref(p), table.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(len(table)), _u32p(tv))
        return tv
