#!/usr/bin/env python3
"""Writes tests/golden/goref/*.npz: inputs and outputs of the REFERENCE'S OWN FUNCTIONS, executed from their Go source text under
/root/reference by tools/go_static/gointerp.py (this image has no Go toolchain; the interpreter knows nothing about TFHE).

    python tools/go_static/make_goref_vectors.py --jobs small                 # seconds .. minutes: FFT, decomposition, rotation, external
                                                                             #   product, CMUX chain, reduced-n bootstrap, lookup tables
    python tools/go_static/make_goref_vectors.py --jobs full --procs 8       # ~25 min per full-size bootstrap: 2 bootstraps, the 10 gates,
                                                                             #   MUX, 3 Uint5 programmable bootstraps, the key ingest

Keys are NOT stored: they are regenerated from a seed by the oracle's harness (tests/oracle_lib.py -- deterministic C PRNG, available
wherever the tests run), handed to the reference code as the Go values it expects, and the fixture keeps the seed, the inputs and what
the reference computed.  The tests (tests/test_goref_vectors.py) then hold the C oracle (CPU tier) and the HIP engine (-m gpu) to those
outputs: bit for bit at the N = 1024, L = 3, Bgbit = 6 sets, by decryption + phase distance at Uint5 (tolerance regime).

Each file records, under "meta", the SHA-256 of every reference source file the interpreter loaded for it, so a fixture names the
source text it was computed from.  Nothing of the reference is copied: the fixtures are numbers.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gointerp as gi  # noqa: E402

REF = os.environ.get("GO_TFHE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "goref")
KEY_SEED_128 = 0x7F4E0002          # = tests/conftest.py keys128: the GPU tier already has this key on the device
KEY_SEED_SMALL = 0x7F4E0003        # = tests/conftest.py keys_small (n = 24)
KEY_SEED_UINT5 = 0x7F4E0091
LEVELS = {"128": "Security128Bit", "80": "Security80Bit", "110": "Security110Bit", "uint5": "SecurityUint5", "uint1": "SecurityUint1", "uint2": "SecurityUint2",
          "uint3": "SecurityUint3", "uint4": "SecurityUint4"}
PARAM_VARS = {"128": "params128Bit", "80": "params80Bit", "110": "params110Bit", "uint5": "paramsUint5", "uint1": "paramsUint1", "uint2": "paramsUint2",
              "uint3": "paramsUint3", "uint4": "paramsUint4"}
KEY_SEED_80 = 0x7F4E0001           # = tests/conftest.py keys80


class LazyRows:
    """A []*tlwe.TLWELv0 whose elements are built from a numpy matrix when the Go code indexes them (the Uint5 key-switching key
    is 393,216 rows of 1,072 words: 1.7 GB as numbers, far more as Go values)."""

    def __init__(self, mat, make):
        self.mat, self.make = mat, make

    def __len__(self):
        return self.mat.shape[0]

    def __getitem__(self, i):
        return self.make(self.mat[i])


class Ref:
    """The reference under the interpreter at one parameter set, plus builders for the Go values its functions take."""

    def __init__(self, level, n_override=None, seed=0x7F4E0050):
        self.I = gi.Interp(REF, seed=seed)
        I = self.I
        self.level = level
        self.pk = {n: I.load(n) for n in ("params", "utils", "poly", "tlwe", "trlwe", "trgsw", "evaluator", "cloudkey", "lut", "key", "gates")}
        I.pkgs[self.pk["params"].path].values["CurrentSecurityLevel"] = I.pkg_value(self.pk["params"], LEVELS[level])
        if n_override is not None:                       # a parameter, not the algorithm: the LWE dimension of this run
            pv = I.pkg_value(self.pk["params"], PARAM_VARS[level])
            pv.f["TLWELv0"].f["N"] = int(n_override)
        self.TORUS = I.named(self.pk["params"], "Torus")
        g = I.call_func("params", "GetTRGSWLv1")
        self.N, self.L = int(g.f["N"]), int(g.f["L"])
        self.n = int(I.call_func("params", "GetTLWELv0").f["N"])
        self.offset = I.call_func("cloudkey", "genDecompositionOffset")
        self.gate_tv = I.call_func("cloudkey", "genTestvec")

    # ---- numpy -> Go values
    def torus(self, a):
        return gi.np_to_slice(np.ascontiguousarray(a, np.uint32), self.TORUS, np.uint32)

    def f64(self, a):
        return gi.np_to_slice(np.ascontiguousarray(a, np.float64), gi.BASIC_RT["float64"], float)

    def lwe(self, row):
        return gi.GoPtr(gi.GoStruct(self.I.named(self.pk["tlwe"], "TLWELv0"), {"P": self.torus(row)}))

    def lwe1(self, row):
        return gi.GoPtr(gi.GoStruct(self.I.named(self.pk["tlwe"], "TLWELv1"), {"P": self.torus(row)}))

    def trlwe(self, ab):
        return gi.GoPtr(gi.GoStruct(self.I.named(self.pk["trlwe"], "TRLWELv1"), {"A": self.torus(ab[0]), "B": self.torus(ab[1])}))

    def trgsw_torus(self, rows):
        els = [self.trlwe(r) for r in rows]
        return gi.GoPtr(gi.GoStruct(self.I.named(self.pk["trgsw"], "TRGSWLv1"), {"TRLWE": gi.GoSlice(els, 0, len(els), len(els), None)}))

    def trgsw_fft_from_arrays(self, rows_f):
        """A trgsw.TRGSWLv1FFT holding exactly the given spectra ([2L][2][N] float64 in the reference's FourierPoly layout)."""
        FP = self.I.named(self.pk["poly"], "FourierPoly")
        RowT = self.I.named(self.pk["trgsw"], "TRLWELv1FFT")
        els = [gi.GoStruct(RowT, {"A": gi.GoStruct(FP, {"Coeffs": self.f64(r[0])}), "B": gi.GoStruct(FP, {"Coeffs": self.f64(r[1])})}) for r in rows_f]
        return gi.GoPtr(gi.GoStruct(self.I.named(self.pk["trgsw"], "TRGSWLv1FFT"), {"TRLWEFFT": gi.GoSlice(els, 0, len(els), len(els), RowT)}))

    def bsk(self, bsk_f):
        els = [self.trgsw_fft_from_arrays(bsk_f[i]) for i in range(bsk_f.shape[0])]
        return gi.GoSlice(els, 0, len(els), len(els), None)

    def ksk(self, ksk):
        rows = LazyRows(ksk, self.lwe)
        return gi.GoSlice(rows, 0, len(rows), len(rows), None)

    def cloudkey(self, bsk_f, ksk):
        return gi.GoPtr(gi.GoStruct(self.I.named(self.pk["cloudkey"], "CloudKey"), {
            "DecompositionOffset": self.offset, "BlindRotateTestvec": self.gate_tv, "KeySwitchingKey": self.ksk(ksk), "BootstrappingKey": self.bsk(bsk_f)}))

    # ---- Go values -> numpy
    @staticmethod
    def u32(s):
        return gi.slice_to_np(s, np.uint32)

    def trlwe_np(self, t):
        return np.stack([self.u32(t.v.f["A"]), self.u32(t.v.f["B"])])

    def new_trlwe(self):
        return self.I.call_func("trlwe", "NewTRLWELv1")

    def new_lwe(self):
        return self.I.call_func("tlwe", "NewTLWELv0")

    def meta(self, what):
        files = {}
        for p in self.I.pkgs.values():
            for ast, _ in p.files:
                rel = os.path.relpath(ast.fname, REF)
                files[rel] = hashlib.sha256(open(ast.fname, "rb").read()).hexdigest()
        return json.dumps({"what": what, "executed_by": "tools/go_static/gointerp.py (a Go-subset interpreter; NOT the Go toolchain)",
                           "reference_files_sha256": files, "statements_executed": self.I.steps, "parameter_set": self.level, "n": self.n})


def oracle():
    from oracle_lib import Oracle
    return Oracle()


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"[goref] wrote {os.path.relpath(path, ROOT)} ({os.path.getsize(path)} bytes)", flush=True)


# ------------------------------------------------------------------------------------------------------------------ small jobs

def job_fft(_):
    o = oracle()
    out = {}
    for N, level in ((1024, "128"), (2048, "uint5")):
        R = Ref(level)
        pe = R.I.call_func("poly", "NewEvaluator", N)
        rs = np.random.RandomState(N)
        Poly, FP = R.I.named(R.pk["poly"], "Poly"), R.I.named(R.pk["poly"], "FourierPoly")
        polys = np.stack([rs.randint(0, 2**32, N, dtype=np.uint64).astype(np.uint32), np.arange(N, dtype=np.uint32) * np.uint32(0x01000193),
                          np.full(N, 0x80000000, np.uint32), np.zeros(N, np.uint32)])
        polys[3, 1] = 1                                   # X
        spectra = []
        for p in polys:
            fp = R.I.call_method(pe, "ToFourierPoly", gi.GoStruct(Poly, {"Coeffs": R.torus(p)}))
            spectra.append(gi.slice_to_np(fp.f["Coeffs"], np.float64))
        spectra = np.stack(spectra)
        # inverse: spectra of small-digit polynomials times a key-like spectrum would need a product; take the forward spectra of
        # SMALL polynomials (|coefficient| < 2^20) scaled by M = N/2 ... simply invert the spectra above (exact round trip expected)
        back = []
        for s in spectra:
            pout = gi.GoStruct(Poly, {"Coeffs": R.torus(np.zeros(N, np.uint32))})
            R.I.call_method(pe, "ToPolyAssignUnsafe", gi.GoStruct(FP, {"Coeffs": R.f64(s.copy())}), pout)
            back.append(R.u32(pout.f["Coeffs"]))
        out[f"polys_{N}"], out[f"spectra_{N}"], out[f"back_{N}"] = polys, spectra, np.stack(back)
        # twiddle tables the reference built (tw, twInv as complex128): the oracle's construction must agree within 2 ulp
        out[f"tw_{N}"] = np.array(pe.v.f["tw"].a[:pe.v.f["tw"].n], np.complex128)
        out["meta"] = R.meta("poly.Evaluator.ToFourierPoly / ToPolyAssignUnsafe (poly/fourier_transform.go:18-44,64-347) at N = 1024 and 2048")
    _ = o
    save("fft", **out)


def job_decompose_rotate(_):
    out = {}
    for level, tag in (("128", "128"), ("uint5", "uint5")):
        R = Ref(level)
        g = R.I.call_func("params", "GetTRGSWLv1")
        bgbit, L, N = int(g.f["BGBIT"]), R.L, R.N
        rs = np.random.RandomState(11 + N)
        p = rs.randint(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
        p[:4] = [0, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF]
        Poly = R.I.named(R.pk["poly"], "Poly")
        outs = [gi.GoStruct(Poly, {"Coeffs": R.torus(np.zeros(N, np.uint32))}) for _ in range(L)]
        R.I.call_func("poly", "DecomposePolyAssign", R.torus(p), bgbit, L, R.offset, gi.GoSlice(outs, 0, L, L, Poly))
        out[f"dec_in_{tag}"], out[f"dec_out_{tag}"] = p, np.stack([R.u32(x.f["Coeffs"]) for x in outs])
        out[f"offset_{tag}"] = np.uint32(R.offset)
        ks = [0, 1, 5, N - 1, N, N + 1, N + 5, 2 * N - 1]
        rot = []
        for k in ks:
            res = R.torus(np.zeros(N, np.uint32))
            R.I.call_func("poly", "PolyMulWithXKInPlace", R.torus(p), int(k), res)
            rot.append(R.u32(res))
        out[f"rot_k_{tag}"], out[f"rot_out_{tag}"] = np.array(ks), np.stack(rot)
        out["meta"] = R.meta("poly.DecomposePolyAssign (poly/decomposer.go:55-66), poly.PolyMulWithXKInPlace (poly/buffer_methods.go:133-164), "
                             "cloudkey.genDecompositionOffset (cloudkey/cloudkey.go:60-71)")
    # utils.F64ToTorus over a grid, and gate constants
    R = Ref("128")
    ds = np.array([0.125, -0.125, 0.25, -0.25, 0.5, -0.5, 0.75, 1.0, 0.0, 1e-9, -1e-9, 0.3, -0.7, 2.0e-5, 3.0e-8])
    out["f64"], out["f64_to_torus"] = ds, np.array([R.I.call_func("utils", "F64ToTorus", float(d)) for d in ds], np.uint32)
    save("decompose_rotate", **out)


def job_other_shapes(_):
    """The transform at N = 512 and decomposition + one external product at every other parameter shape of params.go (Uint1: L = 2, Bgbit = 10;
    Uint2: N = 512, Bgbit = 18; Uint3: Bgbit = 23; Uint4 / Uint5: N = 2048, Bgbit = 22; 80- and 110-bit: the 128-bit ring)."""
    o = oracle()
    out = {}
    R = Ref("uint2")
    pe = R.I.call_func("poly", "NewEvaluator", 512)
    rs = np.random.RandomState(512)
    Poly = R.I.named(R.pk["poly"], "Poly")
    polys = np.stack([rs.randint(0, 2**32, 512, dtype=np.uint64).astype(np.uint32), np.full(512, 0x7FFFFFFF, np.uint32)])
    out["polys_512"] = polys
    out["spectra_512"] = np.stack([gi.slice_to_np(R.I.call_method(pe, "ToFourierPoly", gi.GoStruct(Poly, {"Coeffs": R.torus(p)})).f["Coeffs"], np.float64) for p in polys])
    for level in ("uint1", "uint2", "uint3", "uint4", "80", "110"):
        R = Ref(level, n_override=2)
        g = R.I.call_func("params", "GetTRGSWLv1")
        bgbit, L, N = int(g.f["BGBIT"]), R.L, R.N
        p = o.params(level).small(2)
        rng = o.rng(0x7F4E0200 + N + bgbit)
        s0, s1 = o.keygen_secret(p, rng)
        _, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=False, fourier=True)
        rs = np.random.RandomState(N + bgbit)
        tin = rs.randint(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
        outs = [gi.GoStruct(Poly, {"Coeffs": R.torus(np.zeros(N, np.uint32))}) for _ in range(L)]
        PolyT = R.I.named(R.pk["poly"], "Poly")
        outs = [gi.GoStruct(PolyT, {"Coeffs": R.torus(np.zeros(N, np.uint32))}) for _ in range(L)]
        R.I.call_func("poly", "DecomposePolyAssign", R.torus(tin[0]), bgbit, L, R.offset, gi.GoSlice(outs, 0, L, L, PolyT))
        ev = R.I.call_func("evaluator", "NewEvaluator", N)
        cout = R.new_trlwe()
        R.I.call_method(ev, "ExternalProductAssign", R.trgsw_fft_from_arrays(bsk_f[0]), R.trlwe(tin), R.offset, cout)
        out[f"in_{level}"], out[f"dec_{level}"], out[f"extprod_{level}"] = tin, np.stack([R.u32(x.f["Coeffs"]) for x in outs]), R.trlwe_np(cout)
        out[f"offset_{level}"], out[f"seed_{level}"] = np.uint32(R.offset), np.int64(0x7F4E0200 + N + bgbit)
        out["meta"] = R.meta("poly.Evaluator.ToFourierPoly at N = 512; poly.DecomposePolyAssign + Evaluator.ExternalProductAssign at the Uint1 / Uint2 / Uint3 / Uint4 and "
                             "80- / 110-bit parameter shapes (keys from the oracle harness at n = 2, seed 0x7F4E0200 + N + Bgbit)")
    save("other_shapes", **out)


def job_extprod_chain(_):
    """ExternalProductAssign on one key element, and BlindRotateAssign with the LWE dimension set to 1, 2 and 4 (the accumulator
    after the first CMUX steps), keys_small's first key elements (seed 0x7F4E0003, n = 24)."""
    o = oracle()
    p = o.params("128").small(24)
    rng = o.rng(KEY_SEED_SMALL)
    s0, s1 = o.keygen_secret(p, rng)
    bsk_t, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=True, fourier=True)
    out = {}
    R = Ref("128", n_override=4)
    pe = R.I.call_func("poly", "NewEvaluator", R.N)
    # the reference's own key ingest (trgsw.NewTRGSWLv1FFT, trgsw/trgsw.go:71-82) on the first four key elements
    conv = []
    for i in range(4):
        g = R.I.call_func("trgsw", "NewTRGSWLv1FFT", R.trgsw_torus(bsk_t[i]), pe)
        rows = g.v.f["TRLWEFFT"]
        conv.append(np.stack([np.stack([gi.slice_to_np(rows.a[r].f["A"].f["Coeffs"], np.float64), gi.slice_to_np(rows.a[r].f["B"].f["Coeffs"], np.float64)])
                              for r in range(rows.n)]))
    out["ingest_fourier"] = np.stack(conv)                 # [4][2L][2][N]: must equal the oracle's Fourier key bit for bit
    ev = R.I.call_func("evaluator", "NewEvaluator", R.N)
    rs = np.random.RandomState(7)
    tin = rs.randint(0, 2**32, (2, R.N), dtype=np.uint64).astype(np.uint32)
    cout = R.new_trlwe()
    R.I.call_method(ev, "ExternalProductAssign", R.trgsw_fft_from_arrays(bsk_f[0]), R.trlwe(tin), R.offset, cout)
    out["extprod_in"], out["extprod_out"] = tin, R.trlwe_np(cout)
    # CMuxAssign alone (evaluator/evaluator.go:85-106): out = ct0 + bsk[1] (x) (ct1 - ct0)
    c0 = rs.randint(0, 2**32, (2, R.N), dtype=np.uint64).astype(np.uint32)
    c1 = rs.randint(0, 2**32, (2, R.N), dtype=np.uint64).astype(np.uint32)
    cm = R.new_trlwe()
    R.I.call_method(ev, "CMuxAssign", R.trgsw_fft_from_arrays(bsk_f[1]), R.trlwe(c0), R.trlwe(c1), R.offset, cm)
    out["cmux_ct0"], out["cmux_ct1"], out["cmux_out"] = c0, c1, R.trlwe_np(cm)
    lwe = rs.randint(0, 2**32, 25, dtype=np.uint64).astype(np.uint32)
    lwe[0], lwe[1] = 0, 0xFFFFFFFF                        # the mod-switch edge cases as the first two mask words
    out["chain_lwe"] = lwe
    accs = []
    for K in (1, 2, 4):
        RK = Ref("128", n_override=K)
        evk = RK.I.call_func("evaluator", "NewEvaluator", RK.N)
        ct = np.concatenate([lwe[:K], lwe[-1:]])
        acc = RK.new_trlwe()
        RK.I.call_method(evk, "BlindRotateAssign", RK.lwe(ct), RK.gate_tv, RK.bsk(bsk_f[:K]), RK.offset, acc)
        accs.append(RK.trlwe_np(acc))
    out["chain_acc"] = np.stack(accs)                      # after 1, 2, 4 CMUX steps
    out["meta"] = R.meta("trgsw.NewTRGSWLv1FFT, Evaluator.ExternalProductAssign, CMuxAssign, BlindRotateAssign (evaluator/evaluator.go:50-135) with "
                         "keys_small (tests/conftest.py: 128-bit ring, n = 24, seed 0x7F4E0003); LWE dimension set to 1, 2, 4 for the chain")
    save("extprod_chain_128", **out)


def job_extract_keyswitch(_):
    """trlwe.SampleExtractIndexAssign(., 0, .) (trlwe/trlwe_ops.go:10-21) and trgsw.IdentityKeySwitchingAssign (trgsw/keyswitch.go:10-37) on their own:
    random accumulators (incl. all-zero and all-ones words, whose digits are the k = 0 / k = base-1 edge cases) with keys_small's key-switching key."""
    o = oracle()
    p = o.params("128").small(24)
    rng = o.rng(KEY_SEED_SMALL)
    s0, s1 = o.keygen_secret(p, rng)
    o.keygen_bsk(p, rng, s0, s1, torus=True, fourier=True)
    ksk = o.keygen_ksk(p, rng, s0, s1)
    R = Ref("128", n_override=24)
    rs = np.random.RandomState(31)
    accs = rs.randint(0, 2**32, (3, 2, R.N), dtype=np.uint64).astype(np.uint32)
    accs[1, 0, :8] = 0
    accs[1, 0, 8:16] = 0xFFFFFFFF
    accs[2] = 0
    kskg = R.ksk(ksk)
    exts, outs = [], []
    for acc in accs:
        ext = R.I.call_func("tlwe", "NewTLWELv1")
        R.I.call_func("trlwe", "SampleExtractIndexAssign", R.trlwe(acc), 0, ext)
        out = R.new_lwe()
        R.I.call_func("trgsw", "IdentityKeySwitchingAssign", ext, kskg, out)
        exts.append(R.u32(ext.v.f["P"]))
        outs.append(R.u32(out.v.f["P"]))
    save("extract_keyswitch_n24_128", accs=accs, extracted=np.stack(exts), switched=np.stack(outs),
         meta=R.meta("trlwe.SampleExtractIndexAssign + trgsw.IdentityKeySwitchingAssign with keys_small (128-bit ring, n = 24, seed 0x7F4E0003)"))


def job_small_bootstrap(_):
    """Whole bootstraps and gates at the 128-bit ring with n = 24 (keys_small): BootstrapAssign, every gates.* function and MUX."""
    o = oracle()
    p = o.params("128").small(24)
    rng = o.rng(KEY_SEED_SMALL)
    s0, s1 = o.keygen_secret(p, rng)
    _, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=True, fourier=True)
    ksk = o.keygen_ksk(p, rng, s0, s1)
    R = Ref("128", n_override=24)
    ck = R.cloudkey(bsk_f, ksk)
    erng = o.rng(0x7F4E00A1)
    bits = np.array([[0, 0, 1, 1], [0, 1, 0, 1], [1, 0, 0, 1]])
    a, b, c = (o.encrypt_bools(p, erng, bits[k], s0) for k in range(3))
    out = {"bits": bits.astype(np.uint8), "a": a, "b": b, "c": c}
    ev = R.I.call_func("evaluator", "NewEvaluator", R.N)
    boots = []
    for row in a[:2]:
        res = R.new_lwe()
        R.I.call_method(ev, "BootstrapAssign", R.lwe(row), R.gate_tv, ck.v.f["BootstrappingKey"], ck.v.f["KeySwitchingKey"], R.offset, res)
        boots.append(R.u32(res.v.f["P"]))
    out["bootstrap_out"] = np.stack(boots)
    for name in ("NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN"):
        t0 = time.time()
        out["gate_" + name] = np.stack([R.u32(R.I.call_func("gates", name, R.lwe(a[i]), R.lwe(b[i]), ck).v.f["P"]) for i in range(4)])
        print(f"[goref] gates.{name} x4 at n = 24: {time.time() - t0:.0f} s", flush=True)
    out["gate_MUX"] = np.stack([R.u32(R.I.call_func("gates", "MUX", R.lwe(a[i]), R.lwe(b[i]), R.lwe(c[i]), ck).v.f["P"]) for i in range(4)])
    out["gate_NOT"] = np.stack([R.u32(R.I.call_func("gates", "NOT", R.lwe(a[i])).v.f["P"]) for i in range(4)])
    out["const_true"], out["const_false"] = R.u32(R.I.call_func("gates", "Constant", True).v.f["P"]), R.u32(R.I.call_func("gates", "Constant", False).v.f["P"])
    inputs = gi.GoSlice([gi.GoArray([R.lwe(a[i]), R.lwe(b[i])], 0, 2, None) for i in range(4)], 0, 4, 4, None)
    for name in ("BatchNAND", "BatchAND", "BatchOR", "BatchXOR", "BatchNOR", "BatchXNOR"):
        res = R.I.call_func("gates", name, inputs, ck)
        out["gate_" + name] = np.stack([R.u32(res.a[i].v.f["P"]) for i in range(4)])
    out["meta"] = R.meta("Evaluator.BootstrapAssign (evaluator/evaluator.go:139-148), gates.* and gates.Batch* (gates/gates.go:26-126,156-312) with keys_small "
                         "(128-bit ring, n = 24, seed 0x7F4E0003); inputs from the oracle harness, seed 0x7F4E00A1")
    save("gates_n24_128", **out)


def job_lut(_):
    out = {}
    R = Ref("uint5")
    funcs = {"identity": lambda x: x, "mod16": lambda x: x % 16, "ge16": lambda x: int(x >= 16), "complement": lambda x: 31 - x, "affine": lambda x: (3 * x + 1) % 32}
    gen = R.I.call_func("lut", "NewGenerator", 32)
    for name, f in funcs.items():
        t = R.I.call_func("lut", "NewLookUpTable")
        R.I.call_method(gen, "GenLookUpTableAssign", (lambda a, f=f: int(f(int(a[0])))), t)
        out["uint5_" + name] = R.trlwe_np(t.v.f["Poly"])
    enc = R.I.call_func("lut", "NewEncoder", 32)
    out["uint5_encode"] = np.array([R.I.call_method(enc, "Encode", m) for m in range(-3, 40)], np.uint32)
    out["uint5_encode_in"] = np.arange(-3, 40)
    R2 = Ref("128")
    gen2 = R2.I.call_func("lut", "NewGenerator", 2)
    for name, f in {"id2": lambda x: x, "not2": lambda x: 1 - x}.items():
        t = R2.I.call_func("lut", "NewLookUpTable")
        R2.I.call_method(gen2, "GenLookUpTableAssign", (lambda a, f=f: int(f(int(a[0])))), t)
        out["binary_" + name] = R2.trlwe_np(t.v.f["Poly"])
    out["meta"] = R.meta("lut.Generator.GenLookUpTableAssign (lut/generator.go:56-100), lut.Encoder.Encode (lut/encoder.go:47-74) at Uint5 (modulus 32) and at the "
                         "128-bit ring (modulus 2)")
    save("lut", **out)


def job_refkeygen(_):
    """The reference's OWN key generation and encryption end to end at n = 2: key.NewSecretKey, cloudkey.NewCloudKey, EncryptBool, gates.NAND /
    XOR, DecryptBool -- every random value drawn through the interpreter's math/rand stand-in.  The generated keys are stored (0.6 MB): the
    oracle and the engine must reproduce the reference's gate outputs from the reference's own key material."""
    R = Ref("128", n_override=2, seed=0x7F4E00B7)
    I = R.I
    sk = I.call_func("key", "NewSecretKey")
    t0 = time.time()
    ck = I.call_func("cloudkey", "NewCloudKey", sk)
    print(f"[goref] cloudkey.NewCloudKey at n = 2 under the interpreter: {time.time() - t0:.0f} s", flush=True)
    alpha = float(I.call_func("params", "GetTLWELv0").f["ALPHA"])
    s0 = R.u32(sk.v.f["KeyLv0"])
    s1 = R.u32(sk.v.f["KeyLv1"])
    bskg = ck.v.f["BootstrappingKey"]
    bsk_f = np.stack([np.stack([np.stack([gi.slice_to_np(row.f["A"].f["Coeffs"], np.float64), gi.slice_to_np(row.f["B"].f["Coeffs"], np.float64)])
                                for row in g.v.f["TRLWEFFT"].a]) for g in bskg.a[:bskg.n]])
    kskg = ck.v.f["KeySwitchingKey"]
    ksk = np.stack([R.u32(kskg.a[i].v.f["P"]) for i in range(kskg.n)])
    bits = [(0, 0), (0, 1), (1, 0), (1, 1)]
    A, B, outs, decs = [], [], {"NAND": [], "XOR": []}, {"NAND": [], "XOR": []}
    for x, y in bits:
        ca = I.call_method(I.call_func("tlwe", "NewTLWELv0"), "EncryptBool", bool(x), alpha, sk.v.f["KeyLv0"])
        cb = I.call_method(I.call_func("tlwe", "NewTLWELv0"), "EncryptBool", bool(y), alpha, sk.v.f["KeyLv0"])
        A.append(R.u32(ca.v.f["P"]))
        B.append(R.u32(cb.v.f["P"]))
        for g in ("NAND", "XOR"):
            r = I.call_func("gates", g, ca, cb, ck)
            outs[g].append(R.u32(r.v.f["P"]))
            decs[g].append(bool(I.call_method(r, "DecryptBool", sk.v.f["KeyLv0"])))
    save("refkeygen_n2_128", key_lv0=s0, key_lv1=s1, bsk_fourier=bsk_f, ksk=ksk, a=np.stack(A), b=np.stack(B), bits=np.array(bits, np.uint8),
         gate_NAND=np.stack(outs["NAND"]), gate_XOR=np.stack(outs["XOR"]), dec_NAND=np.array(decs["NAND"]), dec_XOR=np.array(decs["XOR"]),
         offset=np.uint32(R.offset), testvec=R.trlwe_np(R.gate_tv),
         meta=R.meta("key.NewSecretKey, cloudkey.NewCloudKey (cloudkey/cloudkey.go:24-145), TLWELv0.EncryptBool, gates.NAND / XOR, DecryptBool -- the reference's own "
                     "key generation, encryption and decryption at the 128-bit ring with the LWE dimension set to 2; randomness from the interpreter's seeded math/rand"))


def job_reference_tests(_):
    """The reference's OWN unit tests of the slow packages, run under the interpreter with the LWE dimension set to 2 (every test generates
    a cloud key: cloudkey.NewCloudKey at n = 700 is hours of interpretation): gates/gates_test.go and evaluator/programmable_bootstrap_test.go.
    The fast packages (utils, params, lut, poly, tlwe) are run live by tests/test_gointerp_semantics.py."""
    out = {}
    for pkg_name in ("gates", "evaluator"):
        R = Ref("128", n_override=2, seed=0x7F4E00F3)
        R.I.pkg_value(R.pk["params"], "params80Bit").f["TLWELv0"].f["N"] = 2      # programmable_bootstrap_test.go switches to the 80-bit set
        t0 = time.time()
        res = R.I.run_reference_tests(pkg_name)
        out[pkg_name] = {"n_override": 2, "seconds": round(time.time() - t0), "tests": res}
        print(f"[goref] reference tests of {pkg_name}: {sum(1 for v in res.values() if not v['failures'])}/{len(res)} pass, {time.time() - t0:.0f} s", flush=True)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "reference_tests.json"), "w") as fh:
        json.dump({"what": "go-tfhe's own Test* functions executed by tools/go_static/gointerp.py (NOT the Go toolchain), 128-bit ring, LWE dimension set to 2",
                   "packages": out}, fh, indent=1)
    print("[goref] wrote tests/golden/goref/reference_tests.json", flush=True)


def job_reference_tests_uint(_):
    """params/uint_params_test.go -- the reference's own test of the programmable-bootstrap path at the Uint sets (BASELINE config 4's home:
    keygen, lut.Generator, Evaluator.BootstrapLUT through identity / complement / modulo tables, DecryptLWEMessage) -- under the interpreter.
    TestUintParameterProperties runs as it is; TestAllUintParameters with the LWE dimension of Uint1-5 set to 2 (each subtest generates a cloud
    key; Uint6-8 are skipped by the reference itself)."""
    out = {}
    I = gi.Interp(REF, seed=0x7F4E00F4)
    t0 = time.time()
    out["TestUintParameterProperties"] = I.run_reference_tests("params", only={"TestUintParameterProperties"})["TestUintParameterProperties"]
    print(f"[goref] TestUintParameterProperties: {out['TestUintParameterProperties']['failures']} {time.time() - t0:.0f} s", flush=True)
    I = gi.Interp(REF, seed=0x7F4E00F5)
    params = I.load("params")
    for m in range(1, 6):
        I.pkg_value(params, f"paramsUint{m}").f["TLWELv0"].f["N"] = 2
    t0 = time.time()
    res = I.run_reference_tests("params", only={"TestAllUintParameters"})["TestAllUintParameters"]
    res["seconds"] = round(time.time() - t0)
    res["n_override"] = 2
    out["TestAllUintParameters"] = res
    print(f"[goref] TestAllUintParameters: failures {res['failures']} skipped {res['skipped']} {res['statements']} statements {time.time() - t0:.0f} s", flush=True)
    with open(os.path.join(OUT, "reference_tests_uint.json"), "w") as fh:
        json.dump({"what": "params/uint_params_test.go executed by tools/go_static/gointerp.py (NOT the Go toolchain); TestAllUintParameters with the LWE "
                           "dimension of Uint1-5 set to 2", "tests": out}, fh, indent=1)
    print("[goref] wrote tests/golden/goref/reference_tests_uint.json", flush=True)


def job_reference_examples(_):
    """The reference's example PROGRAMS, executed as they are by the interpreter (LWE dimension set to 2: each generates a cloud key), their
    standard output recorded: examples/add_two_numbers (BASELINE config 4: the nibble adder, three Evaluator.BootstrapLUT at the Uint5 ring,
    42 + 137 = 179), examples/simple_gates (every gates.* truth table at the 128-bit ring) and examples/programmable_bootstrap
    (BootstrapFunc / BootstrapLUT through identity, NOT, constants, a reused table and a 2-bit increment at the 80-bit ring)."""
    path = os.path.join(OUT, "reference_examples.json")
    out = json.load(open(path, encoding="utf-8"))["examples"] if os.path.exists(path) else {}
    only = os.environ.get("GOREF_EXAMPLES")                       # comma-separated subset; the others keep their recorded runs
    for name, lower in (("simple_gates", ["params128Bit"]), ("add_two_numbers", ["paramsUint5"]), ("programmable_bootstrap", ["params80Bit"])):
        if only and name not in only.split(","):
            continue
        I = gi.Interp(REF, seed=0x7F4E00F6)
        params = I.load("params")
        for v in lower:
            I.pkg_value(params, v).f["TLWELv0"].f["N"] = 2
        I.stdout = []
        src = os.path.join(REF, "examples", name, "main.go")
        pkg = I.load_source("main", {src: open(src).read()}, path="github.com/thedonutfactory/go-tfhe/examples/" + name)
        t0 = time.time()
        I.call_decl(pkg.funcs["main"], pkg, [], None)
        keep = [l.rstrip("\n") for l in I.stdout
                if any(k in l for k in ("\u2705", "\u274c", "\u2713", "\u2717", "\u2192", "Result", "Expected", "Testing inputs", "expected"))]
        out[name] = {"n_override": 2, "seconds": round(time.time() - t0), "statements": I.steps, "stdout_lines": len(I.stdout), "result_lines": keep}
        print(f"[goref] example {name}: {len(I.stdout)} lines, {time.time() - t0:.0f} s; " + " | ".join(keep[-3:]), flush=True)
    name = "simple_gates_on_the_shim"
    if not only or name in only.split(","):
        # The drop-in claim on the reference's own program: examples/simple_gates with ONE line changed in memory -- the import path of
        # `gates` -> the shim's package ("switch by import path") -- everything else (key.NewSecretKey,
        # cloudkey.NewCloudKey, tlwe encryption, the program text) the reference's.  cgo's "C" is tools/go_static/cmock.py on the oracle.
        import cmock
        MOD = "github.com/thedonutfactory/go-tfhe-gpu"
        I = gi.Interp(REF, seed=0x7F4E00F6)
        I.extra_roots = {MOD: os.path.join(ROOT, "shim", "go")}
        params = I.load("params")
        I.pkg_value(params, "params128Bit").f["TLWELv0"].f["N"] = 2
        mock = cmock.MockC(I, oracle(), device_count=2)
        I.stdout = []
        src = os.path.join(REF, "examples", "simple_gates", "main.go")
        text = open(src).read()
        swapped = text.replace('"github.com/thedonutfactory/go-tfhe/gates"', f'"{MOD}/gates"')
        assert swapped != text and swapped.count(MOD) == 1
        pkg = I.load_source("main", {src: swapped}, path="github.com/thedonutfactory/go-tfhe/examples/simple_gates_gpu")
        t0 = time.time()
        I.call_decl(pkg.funcs["main"], pkg, [], None)
        keep = [l.rstrip("\n") for l in I.stdout if any(k in l for k in ("\u2705", "\u274c", "Testing inputs", "expected"))]
        calls = [c[0] for c in mock.calls]
        out[name] = {"n_override": 2, "seconds": round(time.time() - t0), "stdout_lines": len(I.stdout), "result_lines": keep,
                     "c_abi_calls": {k: calls.count(k) for k in sorted(set(calls))}}
        print(f"[goref] example {name}: {len(I.stdout)} lines, {time.time() - t0:.0f} s; calls {out[name]['c_abi_calls']}; " + " | ".join(keep[-2:]), flush=True)
    name = "add_two_numbers_on_the_shim"
    if not only or name in only.split(","):
        # ... and BASELINE config 4's program: examples/add_two_numbers with the import path of `evaluator` switched to the shim's package
        # (evaluator.NewEvaluator, eval.BootstrapLUT keep the reference's signatures): three programmable bootstraps through the C ABI.
        import cmock
        MOD = "github.com/thedonutfactory/go-tfhe-gpu"
        I = gi.Interp(REF, seed=0x7F4E00F6)
        I.extra_roots = {MOD: os.path.join(ROOT, "shim", "go")}
        params = I.load("params")
        I.pkg_value(params, "paramsUint5").f["TLWELv0"].f["N"] = 2
        mock = cmock.MockC(I, oracle(), device_count=1)
        I.stdout = []
        src = os.path.join(REF, "examples", "add_two_numbers", "main.go")
        text = open(src).read()
        swapped = text.replace('"github.com/thedonutfactory/go-tfhe/evaluator"', f'"{MOD}/evaluator"')
        assert swapped != text and swapped.count(MOD) == 1
        pkg = I.load_source("main", {src: swapped}, path="github.com/thedonutfactory/go-tfhe/examples/add_two_numbers_gpu")
        t0 = time.time()
        I.call_decl(pkg.funcs["main"], pkg, [], None)
        keep = [l.rstrip("\n") for l in I.stdout if any(k in l for k in ("\u2705", "\u274c", "Result", "Expected"))]
        calls = [c[0] for c in mock.calls]
        out[name] = {"n_override": 2, "seconds": round(time.time() - t0), "stdout_lines": len(I.stdout), "result_lines": keep,
                     "c_abi_calls": {k: calls.count(k) for k in sorted(set(calls))}}
        print(f"[goref] example {name}: {len(I.stdout)} lines, {time.time() - t0:.0f} s; calls {out[name]['c_abi_calls']}; " + " | ".join(keep[-3:]), flush=True)
    with open(path, "w", encoding="utf-8") as fh:
        json.dump({"what": "go-tfhe's example programs executed by tools/go_static/gointerp.py (NOT the Go toolchain), LWE dimension set to 2; the lines "
                           "of their standard output that state results", "examples": out}, fh, indent=1, ensure_ascii=False)
    print("[goref] wrote tests/golden/goref/reference_examples.json", flush=True)


def job_shim_go_test(_):
    """shim/go/gates/gates_gpu_test.go -- the Go test a user of the shim is asked to run -- EXECUTED by the interpreter: the shim and the
    reference in one interpreter, cgo's "C" mocked on the oracle (tools/go_static/cmock.py), LWE dimension 1.  Its three Test functions
    compare every gate of the shim WORD FOR WORD with the reference's own and must report no failure."""
    import cmock
    MOD = "github.com/thedonutfactory/go-tfhe-gpu"
    I = gi.Interp(REF, seed=0x7F4E00F7)
    I.extra_roots = {MOD: os.path.join(ROOT, "shim", "go")}
    params = I.load("params")
    I.pkg_value(params, "params128Bit").f["TLWELv0"].f["N"] = 1
    mock = cmock.MockC(I, oracle(), device_count=2)
    pkg = I.pkg_by_import(f"{MOD}/gates")
    t0 = time.time()
    res = I.run_reference_tests("gates", pkg=pkg, directory=os.path.join(ROOT, "shim", "go", "gates"))
    n_gate_ctx = len(mock.ctxs)
    # ... and the seam packages' test (shim/go/trgsw/trgsw_gpu_test.go: trgsw.* and trlwe.SampleExtractIndex against the reference's own)
    tpkg = I.pkg_by_import(f"{MOD}/trgsw")
    res.update(I.run_reference_tests("trgsw", pkg=tpkg, directory=os.path.join(ROOT, "shim", "go", "trgsw")))
    calls = [c[0] for c in mock.calls]
    out = {"what": "shim/go/gates/gates_gpu_test.go executed by tools/go_static/gointerp.py (NOT the Go toolchain) with cgo's C mocked on the CPU oracle, "
                   "LWE dimension 1", "seconds": round(time.time() - t0), "tests": res, "c_abi_calls": {k: calls.count(k) for k in sorted(set(calls))},
           "contexts_created": len(mock.ctxs), "contexts_created_by_the_gate_tests": n_gate_ctx,
           "contexts_alive_at_end": sum(c is not None for c in mock.ctxs)}   # `defer gates.Release(ck)` ran; the seam test keeps the key-less scratch context
    for k, v in res.items():
        print(f"[goref] {k}: failures {v['failures']} ({v['statements']} statements)", flush=True)
    with open(os.path.join(OUT, "shim_go_test_run.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"[goref] wrote tests/golden/goref/shim_go_test_run.json ({time.time() - t0:.0f} s); calls {out['c_abi_calls']}", flush=True)


def job_go_golden_program(_):
    """tools/go_golden/main.go -- the program that pins parity the day someone runs it with the Go toolchain -- EXECUTED by the interpreter
    (os / flag / encoding/binary stand-ins; the LWE dimension of both parameter sets set to 2 so that cloudkey.NewCloudKey is minutes, not
    hours), its .npy files read back by numpy, and tests/test_go_golden.py (CPU tier) run on them: the dump program itself is correct
    (header writer, shapes, the order of the reference calls), not only type-correct.  The files are NOT committed (tests/golden/go/ is
    reserved for what the Go toolchain writes); a summary is."""
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="go_golden_interp_")
    I = gi.Interp(REF, seed=0x7F4E0111)
    params = I.load("params")
    I.pkg_value(params, "params128Bit").f["TLWELv0"].f["N"] = 2
    I.pkg_value(params, "paramsUint5").f["TLWELv0"].f["N"] = 2
    I.flag_overrides = {"out": tmp, "batch": 2, "steps": 2, "uint5": True, "pbs": 3}
    src = os.path.join(ROOT, "tools", "go_golden", "main.go")
    pkg = I.load_source("main", {src: open(src).read()}, path="example.com/go_golden")
    t0 = time.time()
    I.call_decl(pkg.funcs["main"], pkg, [], None)
    secs = time.time() - t0
    files = {}
    for sub in ("small", "big", os.path.join("big", "uint5")):
        d = os.path.join(tmp, sub)
        for f in sorted(os.listdir(d)):
            if f.endswith(".npy"):
                a = np.load(os.path.join(d, f))
                files[os.path.join(sub, f)] = {"dtype": str(a.dtype), "shape": list(a.shape)}
    env = dict(os.environ, TFHE_GO_GOLDEN_SMALL=os.path.join(tmp, "small"), TFHE_GO_GOLDEN_BIG=os.path.join(tmp, "big"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_go_golden.py"), "-q", "-m", "not gpu", "-rA"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("PASSED", "FAILED", "SKIPPED", "ERROR"))]
    rec = {"what": "tools/go_golden/main.go executed by tools/go_static/gointerp.py (NOT the Go toolchain) with the LWE dimension of the 128-bit and Uint5 sets set to 2; "
                   "its output files read back and checked by tests/test_go_golden.py (CPU tier)",
           "main_go_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(), "seconds": round(secs), "statements_executed": I.steps,
           "files_written": files, "pytest_returncode": r.returncode, "pytest_results": lines, "pytest_tail": r.stdout.strip().splitlines()[-1:]}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "go_golden_program_run.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    shutil.rmtree(tmp, ignore_errors=True)
    print(f"[goref] go_golden main.go under the interpreter: {secs:.0f} s, pytest rc {r.returncode}: {rec['pytest_tail']}", flush=True)


# ------------------------------------------------------------------------------------------------------------------ full-size jobs

def full_key_128():
    o = oracle()
    p = o.params("128")
    rng = o.rng(KEY_SEED_128)
    s0, s1 = o.keygen_secret(p, rng)
    bsk_t, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=True, fourier=True)
    ksk = o.keygen_ksk(p, rng, s0, s1)
    return o, p, rng, s0, s1, bsk_t, bsk_f, ksk


def full_inputs_128(o, p, s0):
    erng = o.rng(0x7F4E00C3)
    bits = np.array([[0, 1], [1, 1], [1, 0]])              # two items: (a, b, c) = (0, 1, 1) and (1, 1, 0)
    a, b, c = (o.encrypt_bools(p, erng, bits[k], s0) for k in range(3))
    return bits, a, b, c


def job_full(spec):
    kind, arg = spec
    t0 = time.time()
    if kind in ("boot", "gate"):
        o, p, rng, s0, s1, bsk_t, bsk_f, ksk = full_key_128()
        bits, a, b, c = full_inputs_128(o, p, s0)
        R = Ref("128")
        ck = R.cloudkey(bsk_f, ksk)
        if kind == "boot":
            ev = R.I.call_func("evaluator", "NewEvaluator", R.N)
            acc = R.new_trlwe()
            R.I.call_method(ev, "BlindRotateAssign", R.lwe(a[arg]), R.gate_tv, ck.v.f["BootstrappingKey"], R.offset, acc)
            res = R.new_lwe()
            R.I.call_method(ev, "BootstrapAssign", R.lwe(a[arg]), R.gate_tv, ck.v.f["BootstrappingKey"], ck.v.f["KeySwitchingKey"], R.offset, res)
            save(f"full128_boot{arg}", lwe_in=a[arg], bit=np.uint8(bits[0][arg]), trlwe_acc=R.trlwe_np(acc), lwe_out=R.u32(res.v.f["P"]),
                 meta=R.meta("Evaluator.BlindRotateAssign + BootstrapAssign (evaluator/evaluator.go:110-148) at the FULL 128-bit set (n = 700, N = 1024) with keys128 "
                             "(tests/conftest.py, seed 0x7F4E0002); input from the oracle harness, seed 0x7F4E00C3"))
        else:
            i = 0
            args = [R.lwe(a[i]), R.lwe(b[i])] + ([R.lwe(c[i])] if arg == "MUX" else []) + [ck]
            r = R.I.call_func("gates", arg, *args)
            save(f"full128_gate_{arg}", a=a[i], b=b[i], c=c[i], bits=bits[:, i].astype(np.uint8), out=R.u32(r.v.f["P"]),
                 meta=R.meta(f"gates.{arg} (gates/gates.go) at the FULL 128-bit set (n = 700, N = 1024) with keys128 (seed 0x7F4E0002); inputs from the oracle "
                             "harness, seed 0x7F4E00C3"))
    elif kind == "gate80":
        # BASELINE configs[0]: "single NAND gate, 80-bit params (N = 1024), pure-Go CPU path" -- here it is, from the reference's source, at full size
        o = oracle()
        p = o.params("80")
        rng = o.rng(KEY_SEED_80)
        s0, s1 = o.keygen_secret(p, rng)
        _, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=True, fourier=True)
        ksk = o.keygen_ksk(p, rng, s0, s1)
        erng = o.rng(0x7F4E00C9)
        bits = np.array([[1], [1]])
        a, b = (o.encrypt_bools(p, erng, bits[k], s0) for k in range(2))
        R = Ref("80")
        ck = R.cloudkey(bsk_f, ksk)
        r = R.I.call_func("gates", arg, R.lwe(a[0]), R.lwe(b[0]), ck)
        save(f"full80_gate_{arg}", a=a[0], b=b[0], bits=bits[:, 0].astype(np.uint8), out=R.u32(r.v.f["P"]),
             meta=R.meta(f"gates.{arg} at the FULL 80-bit set (n = 550, N = 1024: BASELINE configs[0]) with keys80 (tests/conftest.py, seed 0x7F4E0001); inputs from the "
                         "oracle harness, seed 0x7F4E00C9"))
    elif kind == "pbsu":
        # one programmable bootstrap at FULL size at each remaining Uint set of params.go (Uint1: N = 1024, L = 2; Uint2: N = 512; Uint3: L = 1, Bgbit = 23;
        # Uint4: N = 2048), message modulus 2 / 4 / 8 / 16, through the table of f(x) = (3x + 1) mod m
        level, m = arg
        o = oracle()
        p = o.params(level)
        seed = 0x7F4E0300 + m
        rng = o.rng(seed)
        s0, s1 = o.keygen_secret(p, rng)
        _, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=False, fourier=True)
        ksk = o.keygen_ksk(p, rng, s0, s1)
        R = Ref(level)
        ck = R.cloudkey(bsk_f, ksk)
        gen = R.I.call_func("lut", "NewGenerator", m)
        table = R.I.call_func("lut", "NewLookUpTable")
        R.I.call_method(gen, "GenLookUpTableAssign", (lambda a_, m=m: (3 * int(a_[0]) + 1) % m), table)
        erng = o.rng(seed + 1)
        msg = m - 1
        ct = o.encrypt_message(p, erng, msg, m, s0)
        ev = R.I.call_func("evaluator", "NewEvaluator", R.N)
        res = R.new_lwe()
        R.I.call_method(ev, "BootstrapLUTAssign", R.lwe(ct), table, ck.v.f["BootstrappingKey"], ck.v.f["KeySwitchingKey"], R.offset, res)
        outp = R.u32(res.v.f["P"])
        save(f"full{level}_pbs", lwe_in=ct, msg=np.int64(msg), modulus=np.int64(m), lut=R.trlwe_np(table.v.f["Poly"]), lwe_out=outp, key_seed=np.int64(seed),
             dec=np.int64(o.decrypt_message(p, m, s0, outp)),
             meta=R.meta(f"Evaluator.BootstrapLUTAssign at the FULL {level} set (n = {p.n}, N = {p.N}) through the table of (3x + 1) mod {m}; key from the oracle harness, "
                         f"seed 0x{seed:X}; tolerance regime for the engine (decryption and phase), same-doubles restatement for the oracle"))
    elif kind == "gate110":
        # the third gate set of params.go (110-bit, "original TFHE reference parameters": n = 630), one gate at full size
        o = oracle()
        p = o.params("110")
        rng = o.rng(0x7F4E0110)
        s0, s1 = o.keygen_secret(p, rng)
        _, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=True, fourier=True)
        ksk = o.keygen_ksk(p, rng, s0, s1)
        erng = o.rng(0x7F4E00CB)
        bits = np.array([[1], [0]])
        a, b = (o.encrypt_bools(p, erng, bits[k], s0) for k in range(2))
        R = Ref("110")
        ck = R.cloudkey(bsk_f, ksk)
        r = R.I.call_func("gates", arg, R.lwe(a[0]), R.lwe(b[0]), ck)
        save(f"full110_gate_{arg}", a=a[0], b=b[0], bits=bits[:, 0].astype(np.uint8), out=R.u32(r.v.f["P"]), key_seed=np.int64(0x7F4E0110),
             meta=R.meta(f"gates.{arg} at the FULL 110-bit set (n = 630, N = 1024) with a key from the oracle harness (seed 0x7F4E0110); inputs seed 0x7F4E00CB"))
    elif kind == "ingest":
        # the reference's key ingest over a whole slice of the full key: trgsw.NewTRGSWLv1FFT(bsk_torus[i]) == the oracle's Fourier key, bit for bit
        o, p, rng, s0, s1, bsk_t, bsk_f, ksk = full_key_128()
        R = Ref("128")
        pe = R.I.call_func("poly", "NewEvaluator", R.N)
        lo, hi = arg
        same = []
        for i in range(lo, hi):
            g = R.I.call_func("trgsw", "NewTRGSWLv1FFT", R.trgsw_torus(bsk_t[i]), pe)
            rows = g.v.f["TRLWEFFT"]
            got = np.stack([np.stack([gi.slice_to_np(rows.a[r].f["A"].f["Coeffs"], np.float64), gi.slice_to_np(rows.a[r].f["B"].f["Coeffs"], np.float64)])
                            for r in range(rows.n)])
            same.append(bool(np.array_equal(got, bsk_f[i])))
        save(f"full128_ingest_{lo}_{hi}", lo=np.int64(lo), hi=np.int64(hi), identical=np.array(same),
             meta=R.meta(f"trgsw.NewTRGSWLv1FFT over key elements [{lo}, {hi}) of keys128: bitwise equality with the oracle's Fourier-domain key recorded per element"))
    elif kind == "pbs":
        o = oracle()
        p = o.params("uint5")
        rng = o.rng(KEY_SEED_UINT5)
        s0, s1 = o.keygen_secret(p, rng)
        _, bsk_f = o.keygen_bsk(p, rng, s0, s1, torus=False, fourier=True)
        ksk = o.keygen_ksk(p, rng, s0, s1)
        R = Ref("uint5")
        ck = R.cloudkey(bsk_f, ksk)
        funcs = [("identity", lambda x: x), ("mod16", lambda x: x % 16), ("ge16", lambda x: int(x >= 16))]
        name, f = funcs[arg]
        gen = R.I.call_func("lut", "NewGenerator", 32)
        table = R.I.call_func("lut", "NewLookUpTable")
        R.I.call_method(gen, "GenLookUpTableAssign", (lambda a_, f=f: int(f(int(a_[0])))), table)
        erng = o.rng(0x7F4E00D5 + arg)
        msg = [5, 27, 16][arg]
        ct = o.encrypt_message(p, erng, msg, 32, s0)
        ev = R.I.call_func("evaluator", "NewEvaluator", R.N)
        res = R.new_lwe()
        R.I.call_method(ev, "BootstrapLUTAssign", R.lwe(ct), table, ck.v.f["BootstrappingKey"], ck.v.f["KeySwitchingKey"], R.offset, res)
        outp = R.u32(res.v.f["P"])
        save(f"fulluint5_pbs_{name}", lwe_in=ct, msg=np.int64(msg), lut=R.trlwe_np(table.v.f["Poly"]), lwe_out=outp,
             dec=np.int64(o.decrypt_message(p, 32, s0, outp)),
             meta=R.meta(f"Evaluator.BootstrapLUTAssign (evaluator/programmable_bootstrap.go:93-115) at the FULL Uint5 set (n = 1071, N = 2048) through the '{name}' table; "
                         f"key from the oracle harness, seed 0x{KEY_SEED_UINT5:X}; tolerance regime (compare by decryption and phase)"))
    print(f"[goref] job {spec}: {time.time() - t0:.0f} s", flush=True)


SMALL = {"fft": job_fft, "decompose_rotate": job_decompose_rotate, "extprod_chain": job_extprod_chain, "lut": job_lut,
         "small_bootstrap": job_small_bootstrap, "refkeygen": job_refkeygen, "reference_tests": job_reference_tests, "other_shapes": job_other_shapes,
         "go_golden_program": job_go_golden_program, "extract_keyswitch": job_extract_keyswitch,
              "reference_tests_uint": job_reference_tests_uint, "reference_examples": job_reference_examples, "shim_go_test": job_shim_go_test}
FULL = [("boot", 0), ("boot", 1)] + [("gate", g) for g in ("NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX")] + \
       [("pbs", 0), ("pbs", 1), ("pbs", 2)] + [("ingest", (i, min(i + 100, 700))) for i in range(0, 700, 100)] + [("gate80", "NAND"), ("gate110", "XOR")] + \
       [("pbsu", ("uint1", 2)), ("pbsu", ("uint2", 4)), ("pbsu", ("uint3", 8)), ("pbsu", ("uint4", 16))]


def run_small(name):
    t0 = time.time()
    SMALL[name](None)
    print(f"[goref] {name}: {time.time() - t0:.0f} s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", default="small", help="small | full | a comma-separated list of small job names | full:<index,...>")
    ap.add_argument("--procs", type=int, default=4)
    args = ap.parse_args()
    if args.jobs == "small":
        work = [(run_small, n) for n in SMALL]
    elif args.jobs == "full":
        work = [(job_full, s) for s in FULL]
    elif args.jobs.startswith("full:"):
        work = [(job_full, FULL[int(i)]) for i in args.jobs[5:].split(",")]
    else:
        work = [(run_small, n) for n in args.jobs.split(",")]
    if args.procs <= 1 or len(work) == 1:
        for fn, a in work:
            fn(a)
        return
    with mp.Pool(args.procs, maxtasksperchild=1) as pool:
        rs = [pool.apply_async(fn, (a,)) for fn, a in work]
        for r in rs:
            r.get()


if __name__ == "__main__":
    main()
