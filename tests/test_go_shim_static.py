"""CPU tier: the Go side of the boundary (shim/go/**) checked WITHOUT a Go toolchain.

The image has no `go`, so the shim cannot be compiled here.  tools/go_static/gocheck.py parses it together with the
top-level declarations of every reference package (read from /root/reference: the test skips where that is absent, e.g. on
the GPU box) and the C header, infers the type of every expression and reports what `go build` would refuse in the places a
cgo shim goes wrong.  This file holds the shim to it, proves the checker is not vacuous (mutations of the shim and round 4's
defective text must be caught, with the expected messages), and keeps INTEGRATION.md's blocks equal to the files."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HDR = os.path.join(ROOT, "include", "tfhe_hip.h")
SHIM = os.path.join(ROOT, "shim", "go")
sys.path.insert(0, os.path.join(ROOT, "tools", "go_static"))

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gates")), reason="/root/reference is absent (GPU box): nothing to resolve the shim against")


def _read(rel):
    with open(os.path.join(ROOT, rel)) as fh:
        return fh.read()


def _check_sources(files):
    """files: {import path: [(fname, src)]} checked as packages of their own (the in-tree shim is not loaded)."""
    import gocheck
    return gocheck.check_shim(REF, HDR, None, extra_sources=files)[0]


def _shim_sources(mutate=None):
    """The five shim packages as extra_sources, optionally with one file's text mutated: {rel path: (old, new)}."""
    mod = "github.com/thedonutfactory/go-tfhe-gpu"
    out = {}
    for d, fname in (("gpu", "gpu.go"), ("gates", "gates_gpu.go"), ("evaluator", "evaluator_gpu.go"), ("trgsw", "trgsw_gpu.go"), ("trlwe", "trlwe_gpu.go")):
        rel = f"shim/go/{d}/{fname}"
        src = _read(rel)
        if mutate and rel in mutate:
            old, new = mutate[rel]
            assert src.count(old) >= 1, f"mutation anchor {old!r} not found in {rel}"
            src = src.replace(old, new, 1)
        out[f"{mod}/{d}"] = [(rel, src)]
    return out


@needs_ref
def test_shim_type_checks_against_the_reference_and_the_c_header():
    import gocheck
    errs, stats = gocheck.check_shim(REF, HDR, SHIM)
    assert not errs, "\n".join(errs)
    assert stats["files"] >= 6 and stats["funcs"] >= 105 and stats["reference_packages"] >= 12, stats


@needs_ref
def test_checker_resolved_the_reference_declarations_the_shim_relies_on():
    # the facts the shim depends on are READ from the reference sources, not assumed: if upstream renames a field or changes a
    # type, this test says so before any Go compiler does
    import gocheck
    w, ref = gocheck.build_world(REF, HDR)
    for p in ref:
        w.resolve_package(p, strict=False)
    mod = "github.com/thedonutfactory/go-tfhe"
    P = lambda name: w.by_path[f"{mod}/{name}"]                                   # noqa: E731
    torus = ("named", f"{mod}/params.Torus")
    assert P("params").types["Torus"] == ("defined", ("basic", "uint32"))          # params/params.go:27: a DEFINED type
    g = dict(P("params").types["TRGSWLv1Params"][1][1])
    assert g["N"] == g["NBIT"] == g["L"] == g["BASEBIT"] == g["IKS_T"] == ("basic", "int")
    assert g["BGBIT"] == g["BG"] == ("basic", "uint32")
    assert dict(P("tlwe").types["TLWELv0"][1][1]) == {"P": ("slice", torus)}
    assert dict(P("trlwe").types["TRLWELv1"][1][1]) == {"A": ("slice", torus), "B": ("slice", torus)}
    assert dict(P("poly").types["FourierPoly"][1][1]) == {"Coeffs": ("slice", ("basic", "float64"))}
    assert dict(P("trgsw").types["TRGSWLv1FFT"][1][1]) == {"TRLWEFFT": ("slice", ("named", f"{mod}/trgsw.TRLWELv1FFT"))}
    fp = ("named", f"{mod}/poly.FourierPoly")
    assert dict(P("trgsw").types["TRLWELv1FFT"][1][1]) == {"A": fp, "B": fp}
    ck = dict(P("cloudkey").types["CloudKey"][1][1])
    assert ck["KeySwitchingKey"] == ("slice", ("ptr", ("named", f"{mod}/tlwe.TLWELv0")))
    assert ck["BootstrappingKey"] == ("slice", ("ptr", ("named", f"{mod}/trgsw.TRGSWLv1FFT")))
    assert ck["DecompositionOffset"] == torus
    assert dict(P("key").types["SecretKey"][1][1]) == {"KeyLv0": ("slice", torus), "KeyLv1": ("slice", torus)}
    assert dict(P("lut").types["LookUpTable"][1][1]) == {"Poly": ("ptr", ("named", f"{mod}/trlwe.TRLWELv1"))}
    assert P("gates").types["Ciphertext"] == ("alias", ("named", f"{mod}/tlwe.TLWELv0"))


@needs_ref
def test_shim_keeps_the_reference_signatures():
    """gates.* (14 scalar + 6 batch, gates/gates.go:26-126,156-312), trgsw.* (trgsw/trgsw.go:108-312, trgsw/keyswitch.go:10),
    trlwe.SampleExtractIndex[Assign] (trlwe/trlwe.go:114, trlwe/trlwe_ops.go:10) and the evaluator's surface on the path
    (evaluator/evaluator.go:50-157, evaluator/programmable_bootstrap.go:16-115, evaluator/gates_helper.go:10-63): every
    function of the reference exists in the shim with an IDENTICAL signature (after alias resolution)."""
    import gocheck
    w, ref = gocheck.build_world(REF, HDR)
    shim = {}
    for d in ("gpu", "gates", "evaluator", "trgsw", "trlwe"):
        shim[d], _ = w.load_package(f"github.com/thedonutfactory/go-tfhe-gpu/{d}", gocheck.read_dir(os.path.join(SHIM, d)), name_hint=d, bodies=False)
    for p in ref:
        w.resolve_package(p, strict=False)
    for p in shim.values():
        w.resolve_package(p)
    assert not w.errors, w.errors
    rg, re_ = w.by_path["github.com/thedonutfactory/go-tfhe/gates"], w.by_path["github.com/thedonutfactory/go-tfhe/evaluator"]
    scalar = ["NAND", "OR", "AND", "XOR", "XNOR", "Constant", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX", "NOT", "Copy"]
    batch = ["BatchNAND", "BatchAND", "BatchOR", "BatchXOR", "BatchNOR", "BatchXNOR"]
    for name in scalar + batch:
        assert name in rg.funcs, f"reference gates.{name} not found"
        assert name in shim["gates"].funcs, f"shim gates.{name} missing"
        assert w.dealias(shim["gates"].funcs[name]) == w.dealias(rg.funcs[name]), \
            f"gates.{name}: shim {gocheck.tstr(w.dealias(shim['gates'].funcs[name]))} vs reference {gocheck.tstr(w.dealias(rg.funcs[name]))}"
    assert w.dealias(("named", "github.com/thedonutfactory/go-tfhe-gpu/gates.Ciphertext")) == ("named", "github.com/thedonutfactory/go-tfhe/tlwe.TLWELv0")
    # SURVEY 8(b) seam 3: the trgsw functions gates.Batch* and the evaluator are built on, and the one trlwe function on the path
    rt, rl = w.by_path["github.com/thedonutfactory/go-tfhe/trgsw"], w.by_path["github.com/thedonutfactory/go-tfhe/trlwe"]
    for name in ("ExternalProductWithFFT", "CMUX", "BlindRotate", "BatchBlindRotate", "IdentityKeySwitching", "IdentityKeySwitchingAssign"):
        assert name in rt.funcs, f"reference trgsw.{name} not found"
        assert name in shim["trgsw"].funcs, f"shim trgsw.{name} missing"
        assert w.dealias(shim["trgsw"].funcs[name]) == w.dealias(rt.funcs[name]), \
            f"trgsw.{name}: shim {gocheck.tstr(w.dealias(shim['trgsw'].funcs[name]))} vs reference {gocheck.tstr(w.dealias(rt.funcs[name]))}"
    for name in ("SampleExtractIndex", "SampleExtractIndexAssign"):
        assert name in rl.funcs and name in shim["trlwe"].funcs, name
        assert w.dealias(shim["trlwe"].funcs[name]) == w.dealias(rl.funcs[name]), f"trlwe.{name}: signature differs"
    assert w.dealias(("named", "github.com/thedonutfactory/go-tfhe-gpu/trgsw.TRGSWLv1FFT")) == ("named", "github.com/thedonutfactory/go-tfhe/trgsw.TRGSWLv1FFT")
    assert w.dealias(("named", "github.com/thedonutfactory/go-tfhe-gpu/trlwe.TRLWELv1")) == ("named", "github.com/thedonutfactory/go-tfhe/trlwe.TRLWELv1")
    methods = ["ExternalProductAssign", "CMuxAssign", "BlindRotateAssign", "BootstrapAssign", "Bootstrap", "BootstrapLUTAssign", "BootstrapLUT", "BootstrapLUTTemp", "BootstrapFunc",
               "BootstrapFuncAssign", "PrepareNAND", "PrepareAND", "PrepareOR", "PrepareXOR"]
    for m in methods:
        assert ("Evaluator", m) in re_.methods, f"reference Evaluator.{m} not found"
        assert ("Evaluator", m) in shim["evaluator"].methods, f"shim Evaluator.{m} missing"
        assert w.dealias(shim["evaluator"].methods[("Evaluator", m)]) == w.dealias(re_.methods[("Evaluator", m)]), f"Evaluator.{m}: signature differs"
    # NewEvaluator / ShallowCopy differ only in the Evaluator type they return
    assert shim["evaluator"].funcs["NewEvaluator"][1] == re_.funcs["NewEvaluator"][1] == (("basic", "int"),)


@needs_ref
def test_round4_shim_text_is_rejected_at_exactly_its_three_torus_crossings():
    src = _read("tests/golden/go_shim_r04_defective.go.txt")
    errs = _check_sources({"example.com/r04/gpu": [("r04_gpu.go", src)]})
    assert len(errs) == 3, errs
    lines = src.splitlines()
    sites = sorted(lines[int(re.search(r":(\d+):", e).group(1)) - 1].strip() for e in errs)
    assert sites == sorted(["ksk = append(ksk, row.P...)", "flat = append(flat, c.P...)",
                            "res[i] = &tlwe.TLWELv0{P: out[i*n1 : (i+1)*n1 : (i+1)*n1]}"]), sites
    assert all("params.Torus" in e and "uint32" in e for e in errs)


MUTATIONS = [
    # (file, old, new, expected message fragment)
    ("shim/go/gpu/gpu.go", "row.A.Coeffs...", "row.A.Coefs...", "has no field or method Coefs"),
    ("shim/go/gpu/gpu.go", "g := params.GetTRGSWLv1()\n\tl0", "g := params.GetTRGSWLv2()\n\tl0", "undefined: params.GetTRGSWLv2"),
    ("shim/go/gpu/gpu.go", "t: C.int32_t(g.IKS_T)", "t: C.int32_t(g.IKST)", "has no field or method IKST"),
    ("shim/go/gpu/gpu.go", "check(C.tfhe_ctx_create(&p, C.int(device), &k.ctx))", "check(C.tfhe_ctx_create(&p, device, &k.ctx))", "cannot use int as C.int"),
    ("shim/go/gpu/gpu.go", "check(C.tfhe_load_ksk(k.ctx, torusPtr(rows)))", "check(C.tfhe_load_ksk(k.ctx, torusPtr(rows), 0))", "argument(s) for 2 parameter(s)"),
    ("shim/go/gpu/gpu.go", "rows := make([]params.Torus, 0, len(ksk)*k.n1)", "rows := make([]uint32, 0, len(ksk)*k.n1)", "in append"),
    ("shim/go/gpu/gpu.go", "check(C.tfhe_key_size(k.ctx, C.int(which), &n))", "check(C.tfhe_key_sizes(k.ctx, C.int(which), &n))", "undefined: C.tfhe_key_sizes"),
    ("shim/go/gpu/gpu.go", "flat := make([]float64, 0, len(bsk)*2*g.L*2*g.N)", "flat := make([]float32, 0, len(bsk)*2*g.L*2*g.N)", "in append"),
    ("shim/go/gpu/gpu.go", "\treturn unflatten(out, k.n1)\n}\n\n// GateBatchOps", "\treturn out\n}\n\n// GateBatchOps", "in return value"),
    ("shim/go/gpu/gpu.go", "\t\"github.com/thedonutfactory/go-tfhe/key\"\n", "\t\"github.com/thedonutfactory/go-tfhe/key\"\n\t\"github.com/thedonutfactory/go-tfhe/utils\"\n", "imported and not used"),
    ("shim/go/gpu/gpu.go", "\tfa := flatten(a, k.n1)\n\tfb := flatten(b, k.n1)\n\tvar fc []params.Torus\n\tif c != nil {\n\t\tfc = flatten(c, k.n1)\n\t}\n\tout := make([]params.Torus, len(fa))\n\tlocked(func() {\n\t\tcheck(C.tfhe_gate_batch(k.ctx, nil,",
     "\tfa := flatten(a, k.n1)\n\tspare := 1\n\tfb := flatten(b, k.n1)\n\tvar fc []params.Torus\n\tif c != nil {\n\t\tfc = flatten(c, k.n1)\n\t}\n\tout := make([]params.Torus, len(fa))\n\tlocked(func() {\n\t\tcheck(C.tfhe_gate_batch(k.ctx, nil,", "spare declared and not used"),
    ("shim/go/gpu/gpu.go", "sk.KeyLv0), torusPtr(sk.KeyLv1)", "sk.KeyLv0), torusPtr(sk.KeyLvl1)", "has no field or method KeyLvl1"),
    ("shim/go/gates/gates_gpu.go", "return gpu.Attached(ck.BootstrappingKey, ck.KeySwitchingKey)", "return gpu.Attached(ck.KeySwitchingKey, ck.BootstrappingKey)", "cannot use"),
    ("shim/go/gates/gates_gpu.go", "return gate(gpu.OpNAND, tlweA, tlweB, ck)", "return gate(gpu.OpNANDS, tlweA, tlweB, ck)", "undefined: gpu.OpNANDS"),
    ("shim/go/gates/gates_gpu.go", "\t\tout.SetB(eighth)\n", "\t\tout.SetB(0.125)\n", "cannot use untyped float"),
    ("shim/go/evaluator/evaluator_gpu.go", "e.BootstrapAssign(ctIn, lut.Poly, bsk, ksk, decompositionOffset, ctOut)", "e.BootstrapAssign(ctIn, lut, bsk, ksk, decompositionOffset, ctOut)", "cannot use"),
    ("shim/go/evaluator/evaluator_gpu.go", "copy(ctOut.P, res[0].P)", "copy(ctOut.P, res[0].A)", "has no field or method A"),
    ("shim/go/trgsw/trgsw_gpu.go", "return gpu.Scratch().CMuxWith(cond, []*trlwe.TRLWELv1{in1}, []*trlwe.TRLWELv1{in2}, decompositionOffset)[0]",
     "return gpu.Scratch().CMuxWith(in1, []*trlwe.TRLWELv1{in1}, []*trlwe.TRLWELv1{in2}, decompositionOffset)[0]", "cannot use"),
    ("shim/go/trgsw/trgsw_gpu.go", "return gpu.AttachedKSK(keySwitchingKey).Pick().KeySwitch([]*tlwe.TLWELv1{src})[0]",
     "return gpu.AttachedKSK(keySwitchingKey).Pick().KeySwitch([]*tlwe.TLWELv0{src})[0]", "cannot use"),
    ("shim/go/trlwe/trlwe_gpu.go", "copy(output.P, SampleExtractIndex(trlwe, k).P)", "copy(output.P, SampleExtractIndex(trlwe, k).A)", "has no field or method A"),
    ("shim/go/gpu/gpu.go", "check(C.tfhe_keyswitch_batch(k.ctx, torusPtr(fin), torusPtr(out), C.int(len(in))))", "check(C.tfhe_keyswitch_batch(k.ctx, torusPtr(fin), torusPtr(out)))", "argument(s) for 4 parameter(s)"),
    ("shim/go/gpu/gpu.go", "C.uint32_t(decompositionOffset), torusPtr(fin), torusPtr(out), C.int(len(in))))", "decompositionOffset, torusPtr(fin), torusPtr(out), C.int(len(in))))", "cannot use"),
    ("shim/go/evaluator/evaluator_gpu.go", "lookupTable := generator.GenLookUpTable(f)\n\treturn", "lookupTable := generator.GenLookupTable(f)\n\treturn", "has no field or method GenLookupTable"),
]


@needs_ref
@pytest.mark.parametrize("case", MUTATIONS, ids=[f"{i}:{m[3][:28]}" for i, m in enumerate(MUTATIONS)])
def test_checker_catches_a_seeded_defect(case):
    rel, old, new, frag = case
    errs = _check_sources(_shim_sources({rel: (old, new)}))
    assert errs, f"mutation {old!r} -> {new!r} in {rel} was not noticed"
    assert any(frag in e for e in errs), (frag, errs)


@needs_ref
def test_go_golden_program_type_checks():
    # tools/go_golden/main.go (the program that will pin parity the day someone has Go) against the reference's declarations:
    # every reference function it calls exists with those argument types.  The standard library is not modelled (lenient).
    import gocheck
    src = _read("tools/go_golden/main.go")
    errs, _ = gocheck.check_shim(REF, HDR, None, extra_sources={"example.com/cmd/go_golden": [("tools/go_golden/main.go", src)]}, lenient_std=True)
    assert not errs, "\n".join(errs)
    for needle in ("gen.GenLookUpTableAssign(f, tables[i])", "eval.BootstrapLUTAssign(ins[i], tables[w], ck.BootstrappingKey, ck.KeySwitchingKey, ck.DecompositionOffset, outs[i])",
                   "eval.BootstrapAssign(", "eval.ExternalProductAssign(", "gates.MUX("):
        assert needle in src, needle
    # and a seeded defect in it is caught
    bad = src.replace("gen.GenLookUpTableAssign(f, tables[i])", "gen.GenLookUpTableAssign(tables[i], f)")
    errs, _ = gocheck.check_shim(REF, HDR, None, extra_sources={"example.com/cmd/go_golden": [("main.go", bad)]}, lenient_std=True)
    assert any("cannot use" in e for e in errs), errs


def test_go_golden_program_has_been_executed_and_its_dump_checked():
    # tools/go_golden/main.go run by the interpreter at a reduced LWE dimension (tools/go_static/make_goref_vectors.py --jobs go_golden_program):
    # every file of its documented schema was written and read back by numpy, and tests/test_go_golden.py passed on them
    import hashlib
    import json
    path = os.path.join(ROOT, "tests", "golden", "goref", "go_golden_program_run.json")
    if not os.path.exists(path):
        pytest.skip("no recorded run of tools/go_golden/main.go under the interpreter")
    rec = json.load(open(path))
    assert rec["main_go_sha256"] == hashlib.sha256(open(os.path.join(ROOT, "tools", "go_golden", "main.go"), "rb").read()).hexdigest(), \
        "tools/go_golden/main.go changed since its recorded run: python tools/go_static/make_goref_vectors.py --jobs go_golden_program"
    assert rec["pytest_returncode"] == 0 and not any(l.startswith(("FAILED", "ERROR")) for l in rec["pytest_results"]), rec["pytest_results"]
    passed = [l for l in rec["pytest_results"] if l.startswith("PASSED")]
    assert len(passed) >= 5, rec["pytest_results"]
    f = rec["files_written"]
    for name in ("small/params.npy", "small/extprod_out.npy", "small/cmux_acc.npy", "big/lwe_out.npy", "big/gate_MUX.npy", "small/uint5_lut_identity.npy",
                 "big/uint5/pbs_out.npy", "big/uint5/ksk.npy"):
        assert name in f, sorted(f)
    assert f["small/extprod_out.npy"] == {"dtype": "uint32", "shape": [2, 1024]} and f["small/uint5_lut_ge16.npy"]["shape"] == [2, 2048]


@needs_ref
def test_go_stubs_declare_what_the_reference_declares():
    """tests/go_stubs/ (the reference's TYPES for the GPU box, where the shim is executed against the real library) against the
    reference: every struct has the same fields with the same types, every function / method the same signature."""
    import gocheck
    w, ref = gocheck.build_world(REF, HDR)
    stub_root = os.path.join(ROOT, "tests", "go_stubs")
    stubs = {}
    for d in sorted(os.listdir(stub_root)):
        if os.path.isdir(os.path.join(stub_root, d)):
            stubs[d], _ = w.load_package(f"stub.example/{d}", gocheck.read_dir(os.path.join(stub_root, d)), name_hint=d, bodies=True)
    # the stubs import each other under the reference's paths: resolve those imports against the STUB packages
    mod = "github.com/thedonutfactory/go-tfhe"
    saved = {f"{mod}/{d}": w.by_path[f"{mod}/{d}"] for d in stubs}
    for p in ref:
        w.resolve_package(p, strict=False)
    def norm(t):                                              # "stub.example/x.T" and the reference's "…/go-tfhe/x.T" name the same thing
        if isinstance(t, tuple):
            return tuple(norm(x) for x in t)
        return t.replace("stub.example/", mod + "/") if isinstance(t, str) else t
    for d in stubs:
        w.by_path[f"{mod}/{d}"] = stubs[d]
    for p in stubs.values():
        w.resolve_package(p)
    for d in stubs:
        w.by_path[f"{mod}/{d}"] = saved[f"{mod}/{d}"]
    assert not w.errors, w.errors
    checked = 0
    for d, sp in stubs.items():
        rp = saved[f"{mod}/{d}"]
        for name, (kind, t) in sp.types.items():
            assert name in rp.types, f"{d}.{name} is not a type of the reference"
            if kind == "defined" and t == ("struct", ()):          # an OPAQUE stub (passed through by pointer only): the reference must have a struct of that name
                assert rp.types[name][0] == "defined" and rp.types[name][1][0] == "struct", f"{d}.{name}: not a struct in the reference"
                checked += 1
                continue
            assert rp.types[name][0] == kind and norm(t) == norm(rp.types[name][1]), f"{d}.{name}: stub {t} vs reference {rp.types[name][1]}"
            checked += 1
        for name, ft in sp.funcs.items():
            assert name in rp.funcs and norm(ft) == norm(rp.funcs[name]), f"{d}.{name}: signature differs from the reference's"
            checked += 1
        for key_, ft in sp.methods.items():
            assert key_ in rp.methods and norm(ft) == norm(rp.methods[key_]), f"{d}.{key_}: signature differs from the reference's"
            checked += 1
    assert checked >= 20, checked


def test_integration_md_names_every_shim_file_and_its_one_block_is_verbatim():
    # INTEGRATION.md points at the shim files (it used to inline three of them: 850 lines kept in sync by hand); the one file it does
    # show must equal the file, and every shim file must be named
    import sync_integration_md as sync
    doc = _read("INTEGRATION.md")
    found = dict(sync.blocks(doc))
    assert found, "INTEGRATION.md shows no shim file"
    for rel, body in found.items():
        assert body == sync.render(rel), f"INTEGRATION.md's block for {rel} differs from the file: run python tools/go_static/sync_integration_md.py"
    for d in sorted(os.listdir(SHIM)):
        for f in sorted(os.listdir(os.path.join(SHIM, d))) if os.path.isdir(os.path.join(SHIM, d)) else []:
            assert f"shim/go/{d}/{f}" in doc or (f in doc and f"shim/go/{d}/" in doc), f"INTEGRATION.md does not mention shim/go/{d}/{f}"


def test_shim_calls_every_entry_point_with_the_declared_argument_count():
    # independent of gocheck (and of /root/reference): every C.tfhe_* call in the shim names a declared entry point and passes the
    # declared number of arguments (the lint round 4 ran on the markdown block, now on the files)
    hdr = open(HDR).read()
    decl = {}
    for m in re.finditer(r"^(?:int|const char \*)\s*\*?(tfhe_\w+)\(([^;]*?)\);", hdr, re.M | re.S):
        args = [a.strip() for a in re.sub(r"\s+", " ", m.group(2)).split(",") if a.strip() and a.strip() != "void"]
        decl[m.group(1)] = len(args)
    go = _read("shim/go/gpu/gpu.go")
    calls = set()
    for m in re.finditer(r"C\.(tfhe_\w+)\(", go):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth:
            depth += go[j] == "("
            depth -= go[j] == ")"
            j += 1
        body, d = go[i:j - 1], 0
        n = 1 if body.strip() else 0
        for ch in body:
            d += ch in "(["
            d -= ch in ")]"
            n += ch == "," and d == 0
        assert name in decl, f"gpu.go calls undeclared {name}"
        assert n == decl[name], f"{name}: shim passes {n} arguments, header declares {decl[name]}"
        calls.add(name)
    assert len(calls) >= 20, sorted(calls)
    for other in ("shim/go/gates/gates_gpu.go", "shim/go/evaluator/evaluator_gpu.go", "shim/go/trgsw/trgsw_gpu.go", "shim/go/trlwe/trlwe_gpu.go"):
        assert 'import "C"' not in _read(other), f"{other} must not touch cgo: package gpu is the only cgo layer"


def test_recorded_run_of_the_shims_own_go_tests():
    """shim/go/gates/gates_gpu_test.go and shim/go/trgsw/trgsw_gpu_test.go -- what a Go user of the shim runs -- executed offline by the interpreter
    (cgo's C mocked on the oracle): no failure in their five Test functions (every gate, and every seam function of trgsw / trlwe, word for word
    against the reference's own), 60 gates through the C ABI, every key context the tests created released by their deferred Release / Detach calls
    (what stays alive is the one key-less scratch context of the seam calls, by design)."""
    import json
    path = os.path.join(ROOT, "tests", "golden", "goref", "shim_go_test_run.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/goref/shim_go_test_run.json not generated (make_goref_vectors.py --jobs shim_go_test)")
    rec = json.load(open(path))
    assert "NOT the Go toolchain" in rec["what"]
    assert sorted(rec["tests"]) == ["TestBatchGates", "TestBlindRotateExtractAndKeySwitchEqualTheReferences", "TestExternalProductAndCMUXWithAFreeStandingOperand",
                                    "TestMUXNotCopyConstant", "TestScalarGatesTruthTablesAndWordParity"]
    for name, t in rec["tests"].items():
        assert t["failures"] == [] and not t["skipped"] and t["statements"] > 10**6, (name, t)
    c = rec["c_abi_calls"]
    assert c["gate_batch"] == 60 and c["load_bsk"] == 4 and c["external_product_with"] == 1 and c["cmux_with"] == 1
    assert c["blind_rotate_batch"] == 3 and c["sample_extract_batch"] == 4 and c["keyswitch_batch"] == 2       # batch of 3 over two devices + one scalar
    assert rec["contexts_created_by_the_gate_tests"] == 6           # per gate test: one upload (serves device 0) + one clone
    assert rec["contexts_alive_at_end"] == 1                        # gpu.Scratch()
