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
    "k_blind_rotate_oct<3, 6>": (16, 33, 256, 0),
    "k_blind_rotate_oct<2, 10>": (16, 25, 256, 0),
    "k_blind_rotate_quad<3, 6, 1, 1, 3>": (18, 33, 512, 0),    # one wave per SIMD
    "k_blind_rotate_quad<1, 23, 1, 1, 1>": (14, 17, 256, 0),
    # the N = 2048 step loop exists twice per kernel (one instance per half-tree h: static hand-over patterns), so twice the
    # 16 key-slice loads and twice the scalar loads.  The instances with phase priorities (s_setprio builtins in the loop, two
    # waves per SIMD) carry their 2 x 15 level-1 twiddle loads as vector loads: measured faster than keeping them scalar
    # with register-tied priorities (kernels_n2048.hpp) -- the counts are pinned so that a further change shows up
    "k_blind_rotate_2048<22, false, 1>": (10, 68, 256, 0),
    "k_blind_rotate_2048<22, true, 1>": (20, 38, 256, 0),
    "k_blind_rotate_2048<22, false, 2>": (5, 68, 256, 0),
    "k_blind_rotate_512<18>": (15, 25, 256, 0),
}


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("asm") / "blind_rotate.s"
    src = os.path.join(ROOT, "go-tfhe_amd", "csrc", "blind_rotate.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                    "-S", "--cuda-device-only", src, "-o", str(out)], check=True, capture_output=True)
    text = open(out).read()
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
