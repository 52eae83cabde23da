"""GPU tier: the Go shim (shim/go/**) EXECUTED AGAINST THE REAL LIBRARY.  The Go-subset interpreter (tools/go_static/gointerp.py) runs the
shim's gates / evaluator / gpu packages; cgo's "C" is bound to libtfhe_hip.so through the Python binding's ctypes layer
(tools/go_static/cmock.py: LibBackend), so every C.tfhe_* call the shim makes -- context creation, the flattened key upload,
tfhe_ctx_clone_to, tfhe_gate_batch, tfhe_bootstrap_batch, tfhe_blind_rotate_batch -- reaches the GPU with exactly the buffers the Go code
built.  The reference's packages the shim imports are the declarations of tests/go_stubs/ here (the GPU box has no /root/reference;
tests/test_go_shim_static.py holds the stubs to the reference).  Results are compared with the CPU oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "go_static"))
MOD = "github.com/thedonutfactory/go-tfhe-gpu"


@pytest.fixture(scope="module", params=["oracle-backend", pytest.param("real-library", marks=pytest.mark.gpu)])
def world(request, pkg, oracle, keys_small):
    """"real-library" (-m gpu): C.tfhe_* -> libtfhe_hip.so.  "oracle-backend" (CPU tier, also where /root/reference is absent): the same shim,
    the same stubs, the entry points on the CPU oracle -- the plumbing of this test without a GPU."""
    import cmock
    import gointerp as gi
    k = keys_small
    I = gi.Interp(os.path.join(ROOT, "tests", "go_stubs"))
    I.extra_roots = {MOD: os.path.join(ROOT, "shim", "go")}
    P = I.load("params")
    I.pkg_value(P, "Lv0").f["N"] = int(k.p.n)                               # keys_small: the 128-bit ring with n = 24
    real = request.param == "real-library"
    mock = cmock.MockC(I, oracle, backend=cmock.LibBackend(pkg, oracle) if real else cmock.OracleBackend(oracle, device_count=1))
    TORUS = I.named(P, "Torus")
    torus = lambda a: gi.np_to_slice(np.ascontiguousarray(a, np.uint32), TORUS, np.uint32)                      # noqa: E731
    f64 = lambda a: gi.np_to_slice(np.ascontiguousarray(a, np.float64), gi.BASIC_RT["float64"], float)          # noqa: E731
    T = {n: I.load(n) for n in ("tlwe", "trlwe", "poly", "trgsw", "cloudkey")}
    lwe = lambda row: gi.GoPtr(gi.GoStruct(I.named(T["tlwe"], "TLWELv0"), {"P": torus(row)}))                    # noqa: E731
    FP, ROW = I.named(T["poly"], "FourierPoly"), I.named(T["trgsw"], "TRLWELv1FFT")
    bsk = [gi.GoPtr(gi.GoStruct(I.named(T["trgsw"], "TRGSWLv1FFT"), {"TRLWEFFT": gi.GoSlice(
        [gi.GoStruct(ROW, {"A": gi.GoStruct(FP, {"Coeffs": f64(r[0])}), "B": gi.GoStruct(FP, {"Coeffs": f64(r[1])})}) for r in k.bsk[i]], 0, 2 * k.p.L, 2 * k.p.L, ROW)}))
        for i in range(k.p.n)]
    ksk = [lwe(r) for r in k.ksk]
    tv = gi.GoPtr(gi.GoStruct(I.named(T["trlwe"], "TRLWELv1"), {"A": torus(k.tv[0]), "B": torus(k.tv[1])}))
    ck = gi.GoPtr(gi.GoStruct(I.named(T["cloudkey"], "CloudKey"), {
        "DecompositionOffset": np.uint32(oracle.offset(k.p)), "BlindRotateTestvec": tv,
        "KeySwitchingKey": gi.GoSlice(ksk, 0, len(ksk), len(ksk), None), "BootstrappingKey": gi.GoSlice(bsk, 0, len(bsk), len(bsk), None)}))
    shim = {n: I.pkg_by_import(f"{MOD}/{n}") for n in ("gpu", "gates", "evaluator", "trgsw", "trlwe")}

    def call(p, fn, *a):
        I.ensure_init(shim[p])
        return I.call_decl(shim[p].funcs[fn], shim[p], list(a), None)
    # two contexts on the one GPU of the box: the registry uploads once and replicates with tfhe_ctx_clone_to (device-to-device here)
    call("gpu", "SetDevices", gi.GoSlice([0, 0], 0, 2, 2, gi.BASIC_RT["int"]))
    words = lambda ct: gi.slice_to_np(ct.v.f["P"], np.uint32)                                                    # noqa: E731
    def trl(arr):                                                             # [2][N] words -> *trlwe.TRLWELv1
        return gi.GoPtr(gi.GoStruct(I.named(T["trlwe"], "TRLWELv1"), {"A": torus(arr[0]), "B": torus(arr[1])}))

    def trl_words(t):
        return np.stack([gi.slice_to_np(t.v.f["A"], np.uint32), gi.slice_to_np(t.v.f["B"], np.uint32)])
    yield dict(I=I, gi=gi, mock=mock, ck=ck, lwe=lwe, call=call, words=words, k=k, o=oracle, bsk=bsk, trl=trl, trl_words=trl_words, T=T, torus=torus)
    call("gates", "Release", ck)


def test_the_shims_gates_drive_the_gpu_and_match_the_oracle(world):
    w = world
    k, o = w["k"], w["o"]
    a, b, c = k.enc([1, 0, 1]), k.enc([1, 1, 0]), k.enc([0, 1, 1])
    for name in ("NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN"):
        got = w["words"](w["call"]("gates", name, w["lwe"](a[0]), w["lwe"](b[0]), w["ck"]))
        assert np.array_equal(got, o.gate(k.p, k.bsk, k.ksk, name, a[0], b[0])), name
    got = w["words"](w["call"]("gates", "MUX", w["lwe"](a[1]), w["lwe"](b[1]), w["lwe"](c[1]), w["ck"]))
    assert np.array_equal(got, o.gate(k.p, k.bsk, k.ksk, "MUX", a[1], b[1], c[1]))
    calls = w["mock"].calls
    names = [x[0] for x in calls]
    assert names.count("load_bsk") == 1 and names.count("load_ksk") == 1 and names.count("clone_to") == 1
    live = [x for x in w["mock"].ctxs if x is not None]
    assert [x["clone_path"] for x in live] == [0, 1]                               # the upload serves the GPU itself; the second context is a device-to-device clone (TFHE_OPT_CLONE_PATH)
    assert sum(1 for x in calls if x[0] == "gate_batch") == 11


def test_the_shims_batch_gates_shard_over_two_contexts_on_the_gpu(world):
    w = world
    gi, k, o = w["gi"], w["k"], w["o"]
    bits_a, bits_b = [0, 0, 1, 1, 1, 0, 1], [0, 1, 0, 1, 1, 1, 0]
    a, b = k.enc(bits_a), k.enc(bits_b)
    inputs = gi.GoSlice([gi.GoArray([w["lwe"](a[i]), w["lwe"](b[i])], 0, 2, None) for i in range(7)], 0, 7, 7, None)
    before = len(w["mock"].calls)
    for name, op in (("BatchNAND", "NAND"), ("BatchXOR", "XOR"), ("BatchXNOR", "XNOR")):
        res = w["call"]("gates", name, inputs, w["ck"])
        want, _ = o.gate_batch(k.p, k.bsk, k.ksk, op, a, b)
        assert np.array_equal(np.stack([w["words"](res.a[i]) for i in range(7)]), want), name
    shards = [x[2] for x in w["mock"].calls[before:] if x[0] == "gate_batch"]
    assert sorted(shards[:2]) == [3, 4]                                             # contiguous shards [0, 3) and [3, 7), one per replica
    assert np.array_equal(k.dec(np.stack([w["words"](res.a[i]) for i in range(7)])), np.array(bits_a) == np.array(bits_b))


def test_the_shims_evaluator_bootstraps_on_the_gpu(world):
    w = world
    I, k, o = w["I"], w["k"], w["o"]
    ckf = w["ck"].v.f
    ev = w["call"]("evaluator", "NewEvaluator", int(k.p.N))
    x, y = w["lwe"](k.enc([1])[0]), w["lwe"](k.enc([1])[0])
    prep = I.call_method(ev, "PrepareNAND", x, y)
    assert np.array_equal(w["words"](prep), o.gate_prepare(k.p, "NAND", w["words"](x), w["words"](y)))
    got = I.call_method(ev, "Bootstrap", prep, ckf["BlindRotateTestvec"], ckf["BootstrappingKey"], ckf["KeySwitchingKey"], ckf["DecompositionOffset"])
    assert np.array_equal(w["words"](got), o.bootstrap(k.p, k.bsk, k.ksk, w["words"](prep), k.tv))
    assert bool(k.dec(w["words"](got)[None])[0]) is False
    acc = I.call_func("trlwe", "NewTRLWELv1")
    I.call_method(ev, "BlindRotateAssign", prep, ckf["BlindRotateTestvec"], ckf["BootstrappingKey"], ckf["DecompositionOffset"], acc)
    want = o.blind_rotate(k.p, k.bsk, w["words"](prep), k.tv)
    gi = w["gi"]
    assert np.array_equal(gi.slice_to_np(acc.v.f["A"], np.uint32), want[0]) and np.array_equal(gi.slice_to_np(acc.v.f["B"], np.uint32), want[1])


def test_the_shims_trgsw_and_trlwe_seams_on_the_gpu(world):
    """SURVEY 8(b) seam 3 through the Go files: trgsw.ExternalProductWithFFT / CMUX (a free-standing TRGSW operand travels with the call),
    BlindRotate / BatchBlindRotate (sharded over the two contexts), trlwe.SampleExtractIndex, trgsw.IdentityKeySwitching[Assign],
    Evaluator.ExternalProductAssign / CMuxAssign -- executed by the interpreter, every C.tfhe_* call into the library, results == oracle."""
    w = world
    I, gi, k, o = w["I"], w["gi"], w["k"], w["o"]
    ckf = w["ck"].v.f
    off, tv = ckf["DecompositionOffset"], ckf["BlindRotateTestvec"]
    rs = np.random.RandomState(81)
    r = lambda: rs.randint(0, 2**32, size=(2, k.p.N), dtype=np.uint64).astype(np.uint32)                        # noqa: E731
    x0, x1 = r(), r()
    gsw = w["bsk"][5]
    got = w["call"]("trgsw", "ExternalProductWithFFT", gsw, w["trl"](x0), off, None)
    assert np.array_equal(w["trl_words"](got), o.external_product(k.p, k.bsk[5], x0))
    got = w["call"]("trgsw", "CMUX", w["trl"](x0), w["trl"](x1), gsw, off, None)
    assert np.array_equal(w["trl_words"](got), o.cmux(k.p, k.bsk[5], x0, x1))
    ev = w["call"]("evaluator", "NewEvaluator", int(k.p.N))
    out = I.call_func("trlwe", "NewTRLWELv1")
    I.call_method(ev, "ExternalProductAssign", gsw, w["trl"](x1), off, out)
    assert np.array_equal(w["trl_words"](out), o.external_product(k.p, k.bsk[5], x1))
    acc = w["trl"](x0.copy())
    I.call_method(ev, "CMuxAssign", gsw, acc, w["trl"](x1), off, acc)                  # ctOut == ct0
    assert np.array_equal(w["trl_words"](acc), o.cmux(k.p, k.bsk[5], x0, x1))
    cts = k.enc([1, 0, 1, 1, 0])
    srcs = gi.GoSlice([w["lwe"](c) for c in cts], 0, 5, 5, None)
    before = len(w["mock"].calls)
    res = w["call"]("trgsw", "BatchBlindRotate", srcs, tv, ckf["BootstrappingKey"], off)
    for i in range(5):
        assert np.array_equal(w["trl_words"](res.a[res.o + i]), o.blind_rotate(k.p, k.bsk, cts[i], k.tv)), i
    assert sorted(x[2] for x in w["mock"].calls[before:] if x[0] == "blind_rotate_batch") == [2, 3]
    one = w["call"]("trgsw", "BlindRotate", w["lwe"](cts[2]), tv, ckf["BootstrappingKey"], off, None)
    assert np.array_equal(w["trl_words"](one), w["trl_words"](res.a[res.o + 2]))
    with pytest.raises(gi.GoPanic, match="decompositionOffset"):
        w["call"]("trgsw", "BlindRotate", w["lwe"](cts[2]), tv, ckf["BootstrappingKey"], np.uint32(3), None)
    a0 = w["trl_words"](res.a[res.o])
    for idx in (0, 9, k.p.N - 1):
        e = w["call"]("trlwe", "SampleExtractIndex", res.a[res.o], idx)
        assert np.array_equal(gi.slice_to_np(e.v.f["P"], np.uint32), o.sample_extract(np.ascontiguousarray(a0), idx)), idx
    ext = w["call"]("trlwe", "SampleExtractIndex", res.a[res.o], 0)
    lv0 = w["call"]("trgsw", "IdentityKeySwitching", ext, ckf["KeySwitchingKey"])
    assert np.array_equal(w["words"](lv0), o.key_switch(k.p, k.ksk, o.sample_extract(np.ascontiguousarray(a0), 0)))
    assert np.array_equal(w["words"](lv0), o.bootstrap(k.p, k.bsk, k.ksk, cts[0], k.tv))                       # == the whole bootstrap
    names = [x[0] for x in w["mock"].calls[before:]]
    assert names.count("load_ksk") == 0 and names.count("load_bsk") == 0          # the key switch ran on the cloud key's own replicas


def test_a_library_error_reaches_go_as_a_panic(world):
    w = world
    with pytest.raises(w["gi"].GoPanic, match="tfhe_hip: .*not present"):
        w["call"]("gpu", "UploadKeys", w["ck"].v.f["BootstrappingKey"], w["ck"].v.f["KeySwitchingKey"], 4096)


@pytest.mark.gpu
def test_the_shims_keygen_save_and_load_on_the_gpu(pkg, oracle, keys_small):
    """gpu.NewCloudKey (cloudkey.NewCloudKey replaced by GPU key generation from the secret key), CloudKey.Save / Load (the engine's key
    blobs) through the shim, executed by the interpreter against the real library: a GPU-generated key bootstraps correctly, and a
    context loaded from the saved blobs returns the same words."""
    import cmock
    import gointerp as gi
    k = keys_small
    I = gi.Interp(os.path.join(ROOT, "tests", "go_stubs"))
    I.extra_roots = {MOD: os.path.join(ROOT, "shim", "go")}
    P = I.load("params")
    I.pkg_value(P, "Lv0").f["N"] = int(k.p.n)
    mock = cmock.MockC(I, oracle, backend=cmock.LibBackend(pkg, oracle))
    TORUS = I.named(P, "Torus")
    torus = lambda a: gi.np_to_slice(np.ascontiguousarray(a, np.uint32), TORUS, np.uint32)                      # noqa: E731
    sk = gi.GoPtr(gi.GoStruct(I.named(I.load("key"), "SecretKey"), {"KeyLv0": torus(k.s0), "KeyLv1": torus(k.s1)}))
    gpu = I.pkg_by_import(f"{MOD}/gpu")
    call = lambda fn, *a: I.call_decl(gpu.funcs[fn], gpu, list(a), None)                                        # noqa: E731
    lwe = lambda row: gi.GoPtr(gi.GoStruct(I.named(I.load("tlwe"), "TLWELv0"), {"P": torus(row)}))              # noqa: E731
    key = call("NewCloudKey", sk, 0)
    bits_a, bits_b = [0, 0, 1, 1], [0, 1, 0, 1]
    a, b = k.enc(bits_a), k.enc(bits_b)
    sa = gi.GoSlice([lwe(r) for r in a], 0, 4, 4, None)
    sb = gi.GoSlice([lwe(r) for r in b], 0, 4, 4, None)
    nand = int(I.pkg_value(gpu, "OpNAND"))
    res = I.call_method(key, "GateBatch", nand, sa, sb, None)
    got = np.stack([gi.slice_to_np(res.a[i].v.f["P"], np.uint32) for i in range(4)])
    assert np.array_equal(k.dec(got), ~(np.array(bits_a, bool) & np.array(bits_b, bool)))          # a GPU-generated key works
    blobs = [I.call_method(key, "Save", w) for w in (0, 1)]
    assert [b_.n for b_ in blobs] == [mock.be.key_size(mock.ctxs[0]["h"], w) for w in (0, 1)]
    other = call("newContext", 0)
    for w in (0, 1):
        I.call_method(other, "Load", w, blobs[w])
    res2 = I.call_method(other, "GateBatch", nand, sa, sb, None)
    got2 = np.stack([gi.slice_to_np(res2.a[i].v.f["P"], np.uint32) for i in range(4)])
    assert np.array_equal(got2, got)                                                               # same key, same words
    with pytest.raises(gi.GoPanic, match="tfhe_hip: "):                                            # the other key's blob is refused, as a panic
        I.call_method(other, "Load", 0, blobs[1])
    I.call_method(key, "Close")
    I.call_method(other, "Close")
    assert [c[0] for c in mock.calls].count("keygen") == 1
