"""CPU tier: the C-ABI library loads and exports every declared symbol; host-side logic."""
import ctypes as C

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(pkg):
    declared = pkg.declared_symbols()
    assert len(declared) >= 20
    lib = pkg.load_library()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing


def test_header_is_plain_c(pkg):
    import os, subprocess, tempfile
    hdr = os.path.join(os.path.dirname(pkg.library_path()), "..", "..", "include", "tfhe_hip.h")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "tfhe_hip.h"\nint main(void){tfhe_params p; (void)p; return TFHE_OK;}\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.dirname(os.path.abspath(hdr)),
                        "-c", src, "-o", os.path.join(d, "t.o")], check=True)


def test_no_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.TfheError):
        pkg.Context(pkg.params.Security128Bit)


def test_param_sets_match_reference(pkg, oracle):
    for name in ("80", "110", "128", "uint5"):
        a, b = pkg.params.BY_NAME[name], oracle.params(name)
        assert all(getattr(a, f) == getattr(b, f) for f in ("n", "N", "Nbit", "L", "Bgbit", "basebit", "t"))


def test_gate_coefficients_match_oracle_prepare(pkg, oracle):
    # Evaluator.Prepare* (gates_helper.go:10-63) host mirror vs the oracle
    p = oracle.params("80")
    rs = np.random.RandomState(1)
    a = rs.randint(0, 2**32, p.n + 1, dtype=np.uint64).astype(np.uint32)
    b = rs.randint(0, 2**32, p.n + 1, dtype=np.uint64).astype(np.uint32)
    ev = pkg.evaluator.Evaluator.__new__(pkg.evaluator.Evaluator)
    for name in ("NAND", "AND", "OR", "XOR"):
        assert np.array_equal(getattr(ev, "Prepare" + name)(a, b), oracle.gate_prepare(p, name, a, b))
    assert np.array_equal(pkg.gates.NOT(a), (0 - a).astype(np.uint32))
    assert pkg.gates.Constant(True, p)[p.n] == 0x20000000 and pkg.gates.Constant(False, p)[p.n] == 0xE0000001


def test_shard_bounds_cover_batch(pkg):
    from go_tfhe_amd.distributed import shard_bounds, shard_sizes
    for total in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert sum(shard_sizes(total, world)) == total


def test_balance_levels_keeps_depth_and_dependencies():
    # host-side scheduler of go-tfhe_amd/circuits.py: any width gives a valid topological levelling of the
    # same gates with the same critical-path length; levels respect the width wherever slack allows
    import __graft_entry__ as graft
    graft.load_package()
    from go_tfhe_amd.circuits import ripple_carry_adder, balance_levels, count_gates
    for bits in (1, 2, 8, 16):
        levels, n_wires, sums, cout = ripple_carry_adder(bits)
        flat = sorted(g for l in levels for g in l)
        for width in (1, 2, 4, 100):
            bal = balance_levels(levels, width)
            assert len(bal) == len(levels)
            assert sorted(g for l in bal for g in l) == flat
            have = set(range(2 * bits))
            for lvl in bal:
                for (op, x, y, z, out) in lvl:
                    assert x in have and y in have and (z is None or z in have)
                have |= {g[4] for g in lvl}
            assert all(w in have for w in sums + [cout])
        assert [len(l) for l in balance_levels(levels, 10**6)][0] == 2 * bits       # no limit: the ASAP levelling
    lv8 = balance_levels(ripple_carry_adder(8)[0], 4)
    assert max(len(l) for l in lv8) <= 4
