"""ctypes binding of include/tfhe_hip.h."""
import ctypes as C
import os
import re

import numpy as np

from .build import LIB_PATH
from .params import Params

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "tfhe_hip.h")

OPS = {"NAND": 0, "AND": 1, "OR": 2, "XOR": 3, "XNOR": 4, "NOR": 5,
       "ANDNY": 6, "ANDYN": 7, "ORNY": 8, "ORYN": 9, "MUX": 10}

_lib = None


class TfheError(RuntimeError):
    """A non-zero return from the C ABI (the reference panics at the same places)."""

    def __init__(self, code, msg):
        super().__init__(f"tfhe_hip error {code}: {msg}")
        self.code = code


def library_path():
    return LIB_PATH


def declared_symbols():
    """Every function include/tfhe_hip.h declares."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_[a-z0-9_]+)\s*\(", text)))


def load_library():
    """Load lib/libtfhe_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TfheError(-3, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(the HIP extension is required; there is no CPU fallback)")
    # One HIP runtime per process: torch ships its own libamdhip64, the library links the system one.  Whichever is
    # loaded first serves both -- but only torch-first works (loading torch after the system runtime leaves it with
    # "No HIP GPUs are available"), and the device-pointer entry points exist to be fed torch tensors.  So torch, when
    # installed, is imported before the library is opened; the host-pointer API needs neither torch nor this.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    u32p, u8p, f64p, vp = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.c_void_p
    lib.tfhe_last_error.restype = C.c_char_p
    if hasattr(lib, "tfhe_build_flavor"):
        lib.tfhe_build_flavor.restype = C.c_char_p
        flavor = lib.tfhe_build_flavor().decode()
        if flavor != "release" and not os.environ.get("TFHE_ALLOW_CONTROL_BUILD"):
            raise TfheError(-1, f"{LIB_PATH} is a '{flavor}' build (a test-only variant that misbehaves on purpose); "
                                "set TFHE_ALLOW_CONTROL_BUILD=1 only in the test that needs it")
    sig = {
        "tfhe_device_count": [C.POINTER(C.c_int)],
        "tfhe_ctx_create": [C.POINTER(Params), C.c_int, C.POINTER(vp)],
        "tfhe_ctx_destroy": [vp],
        "tfhe_ctx_clone_to": [vp, C.c_int, C.POINTER(vp)],
        "tfhe_ctx_params": [vp, C.POINTER(Params)],
        "tfhe_ctx_sync": [vp],
        "tfhe_load_bsk_fourier": [vp, f64p],
        "tfhe_load_bsk_torus": [vp, u32p],
        "tfhe_load_ksk": [vp, u32p],
        "tfhe_keygen_cloud": [vp, u32p, u32p, C.c_double, C.c_double, C.c_uint64],
        "tfhe_ctx_reserve": [vp, C.c_int, C.c_int],
        "tfhe_ctx_reserve_extended": [vp, C.c_int, C.c_int],
        "tfhe_key_size": [vp, C.c_int, C.POINTER(C.c_size_t)],
        "tfhe_key_export_dev": [vp, C.c_int, vp, vp],
        "tfhe_key_import_dev": [vp, C.c_int, vp, C.c_size_t, vp],
        "tfhe_key_export": [vp, C.c_int, vp],
        "tfhe_key_import": [vp, C.c_int, vp, C.c_size_t],
        "tfhe_ctx_set_option": [vp, C.c_int, C.c_int],
        "tfhe_ctx_get_option": [vp, C.c_int, C.POINTER(C.c_int)],
        "tfhe_keygen_cloud_seeded": [vp, u32p, u32p, C.c_double, C.c_double, C.POINTER(C.c_uint64)],
        "tfhe_bootstrap_batch": [vp, u32p, u32p, C.c_int, u32p, C.c_int],
        "tfhe_bootstrap_batch_dev": [vp, vp, vp, C.c_int, vp, C.c_int, vp],
        "tfhe_bootstrap_extended_batch": [vp, u32p, u32p, C.c_int, C.c_int, u32p, C.c_int],
        "tfhe_bootstrap_extended_batch_dev": [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp],
        "tfhe_blind_rotate_batch": [vp, u32p, u32p, C.c_int, u32p, C.c_int, C.c_int],
        "tfhe_blind_rotate_batch_dev": [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp],
        "tfhe_external_product_batch": [vp, C.c_int, u32p, u32p, C.c_int],
        "tfhe_extract_keyswitch_batch": [vp, u32p, u32p, C.c_int],
        "tfhe_ctx_decomposition_offset": [vp, u32p],
        "tfhe_external_product_with": [vp, f64p, C.c_uint32, u32p, u32p, C.c_int],
        "tfhe_cmux_with": [vp, f64p, C.c_uint32, u32p, u32p, u32p, C.c_int],
        "tfhe_sample_extract_batch": [vp, u32p, C.c_int, u32p, C.c_int],
        "tfhe_keyswitch_batch": [vp, u32p, u32p, C.c_int],
        "tfhe_extract_keyswitch_batch_dev": [vp, vp, vp, C.c_int, vp],
        "tfhe_gate_batch": [vp, u8p, C.c_int, u32p, u32p, u32p, u32p, C.c_int],
        "tfhe_gate_batch_dev": [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp],
        "tfhe_to_fourier_batch": [vp, u32p, f64p, C.c_int],
        "tfhe_to_poly_batch": [vp, f64p, u32p, C.c_int],
        "tfhe_last_kernel_ms": [vp, C.c_int, C.POINTER(C.c_float)],
        "tfhe_host_alloc": [C.c_size_t, C.POINTER(vp)],
        "tfhe_host_free": [vp],
        "tfhe_timing_enable": [vp, C.c_int],
        "tfhe_timing_read": [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)],
    }
    for name, args in sig.items():
        if not hasattr(lib, name) and os.environ.get("TFHE_HIP_LIB"):
            continue        # an older experiment build under tools/ab_bench.py; the shipped library must export everything
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


def exported_symbols():
    lib = load_library()
    return [s for s in declared_symbols() if hasattr(lib, s)]


def _u32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def _devptr(t, shape=None, device=None, itemsize=4, what="tensor"):
    """Device pointer of a torch tensor (or a raw int / None).  With `shape`, the tensor is checked against it
    (a mismatch would otherwise be an out-of-bounds device access, not an error): element size (int32/uint32 words,
    or bytes for op codes), exact shape, contiguity, and that it lives on the context's GPU."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{what}: device variants need contiguous GPU tensors")
    if shape is not None:
        if t.element_size() != itemsize or t.dtype.is_floating_point:
            raise ValueError(f"{what}: expected {itemsize}-byte integer elements, got {t.dtype}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{what}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        if device is not None and t.device.index != device:
            raise ValueError(f"{what}: lives on cuda:{t.device.index}, the context on cuda:{device}")
    return C.c_void_p(t.data_ptr())


class PinnedArray:
    """numpy view of a page-locked host buffer from tfhe_host_alloc (fast path of the host-pointer ABI)."""

    def __init__(self, shape, dtype=np.uint32):
        lib = load_library()
        self._lib = lib
        self._p = C.c_void_p()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        rc = lib.tfhe_host_alloc(max(nbytes, 1), C.byref(self._p))
        if rc != 0:
            raise TfheError(rc, lib.tfhe_last_error().decode())
        buf = (C.c_char * max(nbytes, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._p is not None and self._p.value:
            self.array = None
            self._lib.tfhe_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU context = evaluator.Evaluator + the loaded cloud key (single submitter)."""

    def __init__(self, params, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.params = params
        self._check(self._lib.tfhe_ctx_create(C.byref(params), int(device), C.byref(self._h)))
        self.device = int(device)

    def _check(self, rc):
        if rc != 0:
            raise TfheError(rc, self._lib.tfhe_last_error().decode())

    def clone_to(self, device):
        """tfhe_ctx_clone_to: a replica of this context (parameters, dispatch limits, keys) on GPU `device`, the keys copied
        GPU to GPU (hipMemcpyPeerAsync; D2D on the same device; host-staged only when the devices are not peers).
        get_option("clone_path") on the result says which path ran."""
        other = Context.__new__(Context)
        other._lib = self._lib
        other._h = C.c_void_p()
        other.params = self.params
        self._check(self._lib.tfhe_ctx_clone_to(self._h, int(device), C.byref(other._h)))
        other.device = int(device)
        return other

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tfhe_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- keys
    def load_bsk_fourier(self, bsk):
        p = self.params
        bsk = np.ascontiguousarray(bsk, dtype=np.float64)
        if bsk.shape != (p.n, 2 * p.L, 2, p.N):
            raise ValueError(f"bsk shape {bsk.shape}")
        self._check(self._lib.tfhe_load_bsk_fourier(self._h, bsk.ctypes.data_as(C.POINTER(C.c_double))))

    def load_bsk_torus(self, bsk):
        p = self.params
        bsk = _u32(bsk, (p.n, 2 * p.L, 2, p.N))
        self._check(self._lib.tfhe_load_bsk_torus(self._h, _p32(bsk)))

    def load_ksk(self, ksk):
        p = self.params
        ksk = _u32(ksk, (p.ksk_rows, p.n + 1))
        self._check(self._lib.tfhe_load_ksk(self._h, _p32(ksk)))

    def keygen_cloud(self, s0, s1, alpha_lv0, alpha_lv1, seed=None):
        """cloudkey.NewCloudKey on the GPU from the binary secret keys (no key upload).
        seed: None = 128 bits from the OS entropy source (the library calls getrandom); an int < 2**128 = a fixed
        seed, FOR TESTS ONLY: the seed determines every mask and noise sample of the cloud key, so it is secret
        key material (include/tfhe_hip.h)."""
        p = self.params
        s0, s1 = _u32(s0, (p.n,)), _u32(s1, (p.N,))
        if seed is None:
            sp = None
        else:
            seed = int(seed)
            if not 0 <= seed < 1 << 128:
                raise ValueError("seed must be a non-negative integer below 2**128")
            sp = (C.c_uint64 * 2)(seed & (2**64 - 1), seed >> 64)
        self._check(self._lib.tfhe_keygen_cloud_seeded(self._h, _p32(s0), _p32(s1), float(alpha_lv0), float(alpha_lv1), sp))

    # ---- host-pointer entry points
    def _tv(self, tv, B):
        p = self.params
        if tv is None:
            return None, 0
        tv = _u32(tv)
        if tv.shape == (2, p.N):
            return tv, 0
        if tv.shape == (B, 2, p.N):
            return tv, 1
        raise ValueError(f"testvec shape {tv.shape}")

    def bootstrap_batch(self, cts, testvec=None):
        p = self.params
        cts = _u32(cts)
        B = cts.shape[0]
        if cts.shape != (B, p.n + 1):
            raise ValueError(f"ciphertext shape {cts.shape}")
        tv, per = self._tv(testvec, B)
        out = np.empty_like(cts)
        self._check(self._lib.tfhe_bootstrap_batch(self._h, _p32(cts), _p32(tv), per, _p32(out), B))
        return out

    def bootstrap_extended_batch(self, cts, lut):
        """Programmable bootstrap through an extended lookup table: lut [ext][2][N] (shared) or [B][ext][2][N]."""
        p = self.params
        cts = _u32(cts)
        B = cts.shape[0]
        lut = _u32(lut)
        per = 1 if lut.ndim == 4 else 0
        ext = lut.shape[1] if per else lut.shape[0]
        if lut.shape[-2:] != (2, p.N) or (per and lut.shape[0] != B):
            raise ValueError("lut must be [ext][2][N] or [B][ext][2][N]")
        out = np.empty((B, p.n + 1), np.uint32)
        self._check(self._lib.tfhe_bootstrap_extended_batch(self._h, _p32(cts), _p32(lut), per, int(ext), _p32(out), B))
        return out

    def blind_rotate_batch(self, cts, testvec=None, nsteps=-1):
        p = self.params
        cts = _u32(cts)
        B = cts.shape[0]
        if cts.shape != (B, p.n + 1):
            raise ValueError(f"ciphertext shape {cts.shape}")
        tv, per = self._tv(testvec, B)
        out = np.empty((B, 2, p.N), np.uint32)
        self._check(self._lib.tfhe_blind_rotate_batch(self._h, _p32(cts), _p32(tv), per, _p32(out), B, int(nsteps)))
        return out

    def external_product_batch(self, key_index, trlwe):
        p = self.params
        trlwe = _u32(trlwe)
        B = trlwe.shape[0]
        if trlwe.shape != (B, 2, p.N):
            raise ValueError(f"trlwe shape {trlwe.shape}")
        out = np.empty_like(trlwe)
        self._check(self._lib.tfhe_external_product_batch(self._h, int(key_index), _p32(trlwe), _p32(out), B))
        return out

    def decomposition_offset(self):
        """CloudKey.DecompositionOffset (cloudkey.go:60-71) as the context derived it."""
        v = C.c_uint32(0)
        self._check(self._lib.tfhe_ctx_decomposition_offset(self._h, C.byref(v)))
        return int(v.value)

    def _trgsw(self, trgsw):
        p = self.params
        g = np.ascontiguousarray(trgsw, dtype=np.float64)
        if g.size != 2 * p.L * 2 * p.N:
            raise ValueError(f"TRGSW operand holds {g.size} doubles, expected [2L][2][N] = {2 * p.L * 2 * p.N}")
        return g

    def external_product_with(self, trgsw, trlwe, offset=None):
        """trgsw.ExternalProductWithFFT / Evaluator.ExternalProductAssign with any TRGSW operand ([2L][2][N] float64, reference layout)."""
        p = self.params
        g = self._trgsw(trgsw)
        trlwe = _u32(trlwe)
        B = trlwe.shape[0]
        if trlwe.shape != (B, 2, p.N):
            raise ValueError(f"trlwe shape {trlwe.shape}")
        out = np.empty_like(trlwe)
        off = self.decomposition_offset() if offset is None else int(offset)
        self._check(self._lib.tfhe_external_product_with(self._h, g.ctypes.data_as(C.POINTER(C.c_double)), off, _p32(trlwe), _p32(out), B))
        return out

    def cmux_with(self, trgsw, ct0, ct1, offset=None):
        """Evaluator.CMuxAssign(ctCond, ct0, ct1): ct0 + cond (x) (ct1 - ct0)."""
        p = self.params
        g = self._trgsw(trgsw)
        ct0, ct1 = _u32(ct0), _u32(ct1)
        B = ct0.shape[0]
        if ct0.shape != (B, 2, p.N) or ct1.shape != ct0.shape:
            raise ValueError(f"trlwe shapes {ct0.shape} / {ct1.shape}")
        out = np.empty_like(ct0)
        off = self.decomposition_offset() if offset is None else int(offset)
        self._check(self._lib.tfhe_cmux_with(self._h, g.ctypes.data_as(C.POINTER(C.c_double)), off, _p32(ct0), _p32(ct1), _p32(out), B))
        return out

    def sample_extract_batch(self, trlwe, k=0):
        """trlwe.SampleExtractIndex for any index k: [B][2][N] -> [B][N+1]."""
        p = self.params
        trlwe = _u32(trlwe)
        B = trlwe.shape[0]
        if trlwe.shape != (B, 2, p.N):
            raise ValueError(f"trlwe shape {trlwe.shape}")
        out = np.empty((B, p.N + 1), np.uint32)
        self._check(self._lib.tfhe_sample_extract_batch(self._h, _p32(trlwe), int(k), _p32(out), B))
        return out

    def keyswitch_batch(self, lwe1):
        """trgsw.IdentityKeySwitching on extracted samples: [B][N+1] -> [B][n+1]."""
        p = self.params
        lwe1 = _u32(lwe1)
        B = lwe1.shape[0]
        if lwe1.shape != (B, p.N + 1):
            raise ValueError(f"TLWELv1 shape {lwe1.shape}")
        out = np.empty((B, p.n + 1), np.uint32)
        self._check(self._lib.tfhe_keyswitch_batch(self._h, _p32(lwe1), _p32(out), B))
        return out

    def extract_keyswitch_batch(self, trlwe):
        p = self.params
        trlwe = _u32(trlwe)
        B = trlwe.shape[0]
        if trlwe.shape != (B, 2, p.N):
            raise ValueError(f"trlwe shape {trlwe.shape}")
        out = np.empty((B, p.n + 1), np.uint32)
        self._check(self._lib.tfhe_extract_keyswitch_batch(self._h, _p32(trlwe), _p32(out), B))
        return out

    def gate_batch(self, ops, a, b, c=None, out=None):
        p = self.params
        a, b = _u32(a), _u32(b)
        B = a.shape[0]
        if a.shape != (B, p.n + 1) or b.shape != a.shape:
            raise ValueError(f"operand shapes {a.shape} {b.shape}")
        c = _u32(c, a.shape) if c is not None else None
        if out is None:
            out = np.empty_like(a)
        elif out.shape != a.shape or out.dtype != np.uint32 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("bad output array")
        if isinstance(ops, str):
            opp, uni = None, OPS[ops]
        else:
            opa = np.ascontiguousarray(ops, dtype=np.uint8)
            if opa.shape != (B,):
                raise ValueError(f"ops shape {opa.shape}")
            opp, uni = opa.ctypes.data_as(C.POINTER(C.c_uint8)), -1
        self._check(self._lib.tfhe_gate_batch(self._h, opp, uni, _p32(a), _p32(b), _p32(c), _p32(out), B))
        return out

    def to_fourier_batch(self, polys):
        polys = _u32(polys)
        P, N = polys.shape
        out = np.empty((P, N), np.float64)
        self._check(self._lib.tfhe_to_fourier_batch(self._h, _p32(polys), out.ctypes.data_as(C.POINTER(C.c_double)), P))
        return out

    def to_poly_batch(self, spectra):
        spectra = np.ascontiguousarray(spectra, dtype=np.float64)
        P, N = spectra.shape
        out = np.empty((P, N), np.uint32)
        self._check(self._lib.tfhe_to_poly_batch(self._h, spectra.ctypes.data_as(C.POINTER(C.c_double)), _p32(out), P))
        return out

    # ---- device-pointer entry points (torch tensors; stream = torch.cuda.Stream or None)
    @staticmethod
    def _stream(stream):
        if stream is None:
            return None
        return C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))

    def gate_batch_dev(self, ops, a, b, c, out, stream=None):
        B, n1, d = a.shape[0], self.params.n + 1, self.device
        if isinstance(ops, str):
            opp, uni = None, OPS[ops]
        else:
            opp, uni = _devptr(ops, (B,), d, 1, "ops"), -1
        self._check(self._lib.tfhe_gate_batch_dev(self._h, opp, uni, _devptr(a, (B, n1), d, what="a"),
                                                  _devptr(b, (B, n1), d, what="b"), _devptr(c, (B, n1), d, what="c"),
                                                  _devptr(out, (B, n1), d, what="out"), B, self._stream(stream)))

    def _tv_dev(self, testvec, B):
        if testvec is None:
            return None, 0
        N, per = self.params.N, 1 if testvec.dim() == 3 else 0
        return _devptr(testvec, (B, 2, N) if per else (2, N), self.device, what="testvec"), per

    def bootstrap_batch_dev(self, cts, testvec, out, stream=None):
        B, n1, d = cts.shape[0], self.params.n + 1, self.device
        tvp, per = self._tv_dev(testvec, B)
        self._check(self._lib.tfhe_bootstrap_batch_dev(self._h, _devptr(cts, (B, n1), d, what="cts"), tvp, per,
                                                       _devptr(out, (B, n1), d, what="out"), B, self._stream(stream)))

    def bootstrap_extended_batch_dev(self, cts, lut, out, stream=None):
        """tfhe_bootstrap_extended_batch_dev: lut = GPU tensor [ext][2][N] (shared) or [B][ext][2][N]."""
        B, n1, d, N = cts.shape[0], self.params.n + 1, self.device, self.params.N
        per = 1 if lut.dim() == 4 else 0
        ext = lut.shape[1] if per else lut.shape[0]
        shape = (B, ext, 2, N) if per else (ext, 2, N)
        self._check(self._lib.tfhe_bootstrap_extended_batch_dev(self._h, _devptr(cts, (B, n1), d, what="cts"),
                                                                _devptr(lut, shape, d, what="lut"), per, int(ext),
                                                                _devptr(out, (B, n1), d, what="out"), B, self._stream(stream)))

    def blind_rotate_batch_dev(self, cts, testvec, out, nsteps=-1, stream=None):
        B, n1, d = cts.shape[0], self.params.n + 1, self.device
        tvp, per = self._tv_dev(testvec, B)
        self._check(self._lib.tfhe_blind_rotate_batch_dev(self._h, _devptr(cts, (B, n1), d, what="cts"), tvp, per,
                                                          _devptr(out, (B, 2, self.params.N), d, what="out"),
                                                          B, int(nsteps), self._stream(stream)))

    def extract_keyswitch_batch_dev(self, trlwe, out, stream=None):
        B, d = trlwe.shape[0], self.device
        self._check(self._lib.tfhe_extract_keyswitch_batch_dev(self._h, _devptr(trlwe, (B, 2, self.params.N), d, what="trlwe"),
                                                               _devptr(out, (B, self.params.n + 1), d, what="out"), B,
                                                               self._stream(stream)))

    def sync(self):
        """Wait for the context's work and report op codes the device rejected (tfhe_ctx_sync)."""
        self._check(self._lib.tfhe_ctx_sync(self._h))

    def key_size(self, which):
        n = C.c_size_t()
        self._check(self._lib.tfhe_key_size(self._h, int(which), C.byref(n)))
        return n.value

    def key_export(self, which):
        """The loaded key `which` as an opaque numpy uint8 array (host memory; save it, ship it, key_import it)."""
        blob = np.empty(self.key_size(which), np.uint8)
        self._check(self._lib.tfhe_key_export(self._h, int(which), blob.ctypes.data_as(C.c_void_p)))
        return blob

    def key_import(self, which, blob):
        """Install a blob from key_export; the library checks its header (parameter set, key kind, layout version,
        length) and raises TfheError when it does not belong to this context."""
        blob = np.ascontiguousarray(blob, np.uint8)
        self._check(self._lib.tfhe_key_import(self._h, int(which), blob.ctypes.data_as(C.c_void_p), blob.size))

    def key_export_dev(self, which, stream=None):
        """The loaded key `which` (0 = bootstrapping, 1 = key-switching) as an opaque uint8 GPU tensor."""
        import torch
        blob = torch.empty(self.key_size(which), dtype=torch.uint8, device=torch.device("cuda", self.device))
        self._check(self._lib.tfhe_key_export_dev(self._h, int(which), C.c_void_p(blob.data_ptr()), self._stream(stream)))
        return blob

    def key_import_dev(self, which, blob, stream=None):
        if not blob.is_cuda or not blob.is_contiguous() or blob.device.index != self.device:
            raise ValueError("key blob: need a contiguous tensor on the context's GPU")
        self._check(self._lib.tfhe_key_import_dev(self._h, int(which), C.c_void_p(blob.data_ptr()),
                                                  blob.numel() * blob.element_size(), self._stream(stream)))

    OPTIONS = {"quad_max": 1, "oct_max": 2, "ks_mfma_min": 3, "frozen": 4, "combine_max": 5, "combine_launches": 6,
               "combine_requests": 7, "ks_wide_ct": 8, "clone_path": 9, "clone_force_host": 10,
               "combine_us_idle": 11, "combine_us_gather": 12, "combine_us_launch": 13, "combine_quiet_us": 14}
    CLONE_PATHS = {0: "not a clone", 1: "same device (D2D)", 2: "peer copy (xGMI)", 3: "host-staged (no peer access)"}

    def set_option(self, name, value):
        """tfhe_ctx_set_option: kernel-dispatch limits for measurements and tests ("quad_max", "oct_max", "ks_mfma_min";
        a negative value restores the default), the "frozen" flag, and "combine_max" -- the largest gate_batch call that is
        combined with concurrent callers' (0 = never; "combine_launches" / "combine_requests" are read-only counters)
        (include/tfhe_hip.h)."""
        self._check(self._lib.tfhe_ctx_set_option(self._h, self.OPTIONS[name], int(value)))

    def get_option(self, name):
        v = C.c_int()
        self._check(self._lib.tfhe_ctx_get_option(self._h, self.OPTIONS[name], C.byref(v)))
        return v.value

    def reserve(self, max_batch, with_mux=False):
        """Pre-size the intermediate buffers (needed before capturing _dev calls into a graph)."""
        self._check(self._lib.tfhe_ctx_reserve(self._h, int(max_batch), int(bool(with_mux))))

    def reserve_extended(self, max_batch, ext):
        """The same for bootstrap_extended_batch_dev with polyExtendFactor `ext` (its accumulators are not sized by reserve)."""
        self._check(self._lib.tfhe_ctx_reserve_extended(self._h, int(max_batch), int(ext)))

    def timing_enable(self, on=True):
        self._check(self._lib.tfhe_timing_enable(self._h, int(bool(on))))

    def timing_read(self, which=0):
        """(launches, total_ms) of kernel `which` since the last read (blocks until finished)."""
        cnt, ms = C.c_int(), C.c_float()
        self._check(self._lib.tfhe_timing_read(self._h, int(which), C.byref(cnt), C.byref(ms)))
        return cnt.value, ms.value

    def last_kernel_ms(self, which=0):
        ms = C.c_float()
        self._check(self._lib.tfhe_last_kernel_ms(self._h, int(which), C.byref(ms)))
        return ms.value
