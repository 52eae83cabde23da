"""CPU tier: properties of the COMPILED blind-rotate kernels that the compiler can silently take away (hipcc cross-compiles
gfx950 without a GPU; ~10 s).

1. Wave-uniform twiddle loads inside the CMUX loops must be scalar loads (s_load).  LLVM only selects them when its
   MemorySSA walk proves the memory unclobbered, and three harmless-looking constructs defeat that walk -- each cost 2-10 %
   of a kernel until found: a global atomic ahead of the loop (round 2), an `asm volatile("")`
   optimisation barrier in a kernel body (round 3: k_blind_rotate_2048, k_blind_rotate_quad), and a uniform-but-conditional
   block that ends in a plain LDS store (round 3: k_blind_rotate_oct).  The symptom is only visible in the instruction mix:
   the kernel's vector loads (global_load_dwordx4) exceed its key-slice and set-up loads.
2. Register and scratch budgets the launch shapes rely on: <= 256 VGPRs (two waves per SIMD), no scratch inside the loops
   of the kernels BASELINE's configs run."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernel (demangled prefix) -> (min s_load_dwordx*, max global_load_dwordx4, max VGPRs, max scratch bytes)
# vector loads = key slices of one step (2 x 8 per gadget level for the two-wave kernels, ...) + the per-lane twiddle set-up
EXPECT = {
    "k_blind_rotate<3, 6, 2, true>": (15, 54, 256, 8),    # headline kernel (full launches); 8 bytes of scratch are outside the loop
    "k_blind_rotate<3, 6, 2, false>": (15, 54, 256, 0),
    "k_blind_rotate<3, 6, 1, false>": (15, 54, 256, 20),
    "k_blind_rotate<3, 6, 1, true>": (15, 54, 256, 24),
    "k_blind_rotate<2, 10, 2, true>": (15, 38, 256, 0),
    "k_blind_rotate<1, 23, 2, true>": (15, 22, 256, 0),
    # eight-wave kernel: a step's key slices are requested in the idle part of the previous step (round 4), so the 16 + 8 (8 + 8)
    # key loads appear twice -- once ahead of the loop for step 0, once inside it -- next to the 9 per-lane twiddle set-up loads
    # (its eight wave-uniform level-1 twiddles are loaded ONCE into SGPRs ahead of the loop since the forward phase carries an
    # s_setprio: three wide s_loads instead of several per step -- the vector-load bound is what catches a regression here)
    "k_blind_rotate_oct<3, 6>": (12, 49, 256, 0),
    "k_blind_rotate_oct<2, 10>": (12, 33, 256, 0),
    "k_blind_rotate_quad<3, 6, 1, 1, 3>": (18, 33, 512, 0),    # one wave per SIMD
    "k_blind_rotate_quad<1, 23, 1, 1, 1>": (14, 17, 256, 0),
    # the N = 2048 step loop exists twice per kernel (one instance per half-tree h: static hand-over patterns), so twice the
    # 16 key-slice loads and twice the scalar loads.  The instances with phase priorities (s_setprio builtins in the loop, two
    # waves per SIMD) carry their 2 x 15 level-1 twiddle loads as vector loads: measured faster than keeping them scalar
    # with register-tied priorities (kernels_n2048.hpp) -- the counts are pinned so that a further change shows up
    # (round 4: the two-workgroups-per-CU instance keeps its 16 level-1 twiddles in SGPRs, loaded once ahead of the loops: its vector
    # loads are the 2 x 16 key slices + set-up again)
    "k_blind_rotate_2048<22, false, 1>": (5, 40, 256, 0),
    "k_blind_rotate_2048<22, true, 1>": (20, 38, 256, 0),
    "k_blind_rotate_2048<22, false, 2>": (5, 68, 256, 0),
    "k_blind_rotate_512<18>": (15, 25, 256, 0),
}


def step_loop_mix(body):
    """Static instruction mix of the kernel's CMUX step loop: the innermost-numbered backward branch region that contains an
    s_barrier and the most instructions.  Returns {"valu": n, "ds": {opcode: n}, "barriers": n} (None when no such loop)."""
    lines = body.splitlines()
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if not m or labels.get(m.group(1), len(lines)) >= i:
            continue
        ops = [x.split()[0] for x in lines[labels[m.group(1)]:i + 1] if re.match(r"^\s+[a-z]", x)]
        if "s_barrier" not in ops:
            continue
        if best is None or len(ops) > len(best):
            best = ops
    if best is None:
        return None
    ds = {}
    for x in best:
        if x.startswith("ds_"):
            ds[x] = ds.get(x, 0) + 1
    return {"valu": sum(1 for x in best if x.startswith("v_")), "ds": ds, "barriers": best.count("s_barrier")}


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    # the blind-rotate translation units, each with the machine-scheduler options the build gives it (go-tfhe_amd/build.py)
    import importlib.util
    spec = importlib.util.spec_from_file_location("tfhe_build", os.path.join(ROOT, "go-tfhe_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    units = [(src, extra) for src, extra in bld.SOURCES if src.startswith("blind_rotate")]
    assert len(units) == 3, units
    tmp = tmp_path_factory.mktemp("asm")
    procs = []
    for src, extra in units:
        out = tmp / (src + ".s")
        procs.append((out, subprocess.Popen([HIPCC] + bld.FLAGS + extra + ["-S", "--cuda-device-only", os.path.join(bld.CSRC, src), "-o", str(out)],
                                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    text = ""
    for out, pr in procs:
        assert pr.wait() == 0, out
        text += open(out).read() + "\n"
    dem = subprocess.run(["c++filt"], input="\n".join(sorted(set(re.findall(r"^(_ZN4tfhe\w+):", text, re.M)))),
                         capture_output=True, text=True, check=True).stdout.splitlines()
    names = dict(zip(sorted(set(re.findall(r"^(_ZN4tfhe\w+):", text, re.M))), dem))
    kernels = {}
    for sym, pretty in names.items():
        body = text[text.index(f"\n{sym}:"):]
        body = body[: body.index("s_endpgm")]
        meta = text[text.index(f".amdhsa_kernel {sym}\n"):]
        meta = meta[: meta.index(".end_amdhsa_kernel")]
        key = re.sub(r"^void tfhe::|\(tfhe::BlindRotateArgs\)$", "", pretty)
        kernels[key] = {
            "loop": step_loop_mix(body),
            "s_load": len(re.findall(r"^\s*s_load_dwordx", body, re.M)),
            "v_load": len(re.findall(r"^\s*global_load_dwordx4", body, re.M)),
            "vgpr": int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)),
            "scratch": int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)),
        }
    return kernels


@pytest.mark.parametrize("kernel", sorted(EXPECT))
def test_uniform_twiddle_loads_are_scalar_and_budgets_hold(asm, kernel):
    assert kernel in asm, f"{kernel} is not instantiated any more; have {sorted(asm)}"
    k, (min_s, max_v, max_vgpr, max_scratch) = asm[kernel], EXPECT[kernel]
    assert k["v_load"] <= max_v, (f"{kernel}: {k['v_load']} vector loads > {max_v}: wave-uniform twiddle loads of the CMUX loop have "
                                  "stopped being scalar loads (see this file's docstring)")
    assert k["s_load"] >= min_s, f"{kernel}: only {k['s_load']} scalar loads (expected >= {min_s})"
    assert k["vgpr"] <= max_vgpr, f"{kernel}: {k['vgpr']} VGPRs"
    assert k["scratch"] <= max_scratch, f"{kernel}: {k['scratch']} bytes of scratch (spills)"


def test_bench_constants_equal_the_compiled_step_loop(asm):
    """bench.py's `roofline.attainable` and `lds_pipe` are computed from the headline kernel's instruction counts per wave and
    CMUX step.  They are constants in bench.py -- and this test holds them to the compiled kernel: the static VALU count of
    the step loop (+-2: an instruction moved across the loop boundary is not a change of the kernel's cost) and every DS opcode
    count exactly.  (The static counts equal the PMC counters of the committed rocprofv3 pass -- SQ_INSTS_VALU 2.1437e9 /
    2,048 waves / 700 steps = 1,495, SQ_INSTS_LDS 2.667e8 -> 186 -- because the step loop is one straight-line block.)"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.BR_KERNEL in asm, f"{bench.BR_KERNEL} is not instantiated any more"
    mix = asm[bench.BR_KERNEL]["loop"]
    assert mix is not None and mix["barriers"] == 1, mix
    assert abs(mix["valu"] - bench.BR_VALU_PER_WAVE_STEP) <= 2, \
        f"step loop of {bench.BR_KERNEL} has {mix['valu']} VALU instructions, bench.BR_VALU_PER_WAVE_STEP says {bench.BR_VALU_PER_WAVE_STEP}"
    assert mix["ds"] == bench.BR_DS_PER_WAVE_STEP, f"DS instructions of the step loop {mix['ds']} != bench.BR_DS_PER_WAVE_STEP {bench.BR_DS_PER_WAVE_STEP}"
    for kind in bench.BR_DS_PER_WAVE_STEP:                     # every kind has a price in the LDS-pipe model
        base, _ = bench.BR_DS_PRICED_AS.get(kind, (kind, 1.0))
        assert base in ("ds_write_b128", "ds_read_b128", "ds_read_b32", "ds_add_u32"), kind


def test_traffic_source_names_the_committed_measurement():
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    src = bench.traffic_source()
    assert src["file"] == "profiles/pmc_traffic.json" and os.path.exists(os.path.join(ROOT, src["file"]))
    assert src["collected"] and "source_commit" in src["collected"]
    for f in ("pmc_traffic.json", "pmc_traffic_uint5.json"):
        rec = json.load(open(os.path.join(ROOT, "profiles", f)))
        for k in ("k_blind_rotate", "k_keyswitch"):
            assert rec[k]["hbm_bytes_per_launch"] == pytest.approx((2 * rec[k]["FETCH_SIZE_KiB"] + rec[k]["WRITE_SIZE_KiB"]) * 1024)
