"""GPU tier: the C ABI under misuse -- every entry point called straight through ctypes with a NULL context, with NULL buffers on a live
context, and with negative counts (tests/abi_misuse_sweep.py, in a subprocess so that a crash is a test failure and not the end of the
session).  include/tfhe_hip.h: every function returns 0 or a negative code, never throws, and leaves a message in tfhe_last_error."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_every_entry_point_refuses_null_and_negative_arguments_without_crashing(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_misuse_sweep.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"the sweep died with {r.returncode} (a crash inside an entry point?)\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
    line = [l for l in r.stdout.splitlines() if l.startswith("SWEEP ")][-1]
    res = json.loads(line[6:])
    assert res.pop("_context_still_works") is True
    bad = {}
    for name, row in res.items():
        if name == "tfhe_ctx_destroy":
            assert row["null_ctx"] == 0                       # destroying nothing is fine
            continue
        if name == "tfhe_host_free":
            continue
        if "null_ctx" in row and not (row["null_ctx"] < 0 and row["null_ctx_msg"]):
            bad[name + ":null_ctx"] = row
        if isinstance(row.get("null_args"), int) and row["null_args"] >= 0:
            bad[name + ":null_args"] = row
        if "negative_counts" in row and row["negative_counts"] >= 0:
            bad[name + ":negative_counts"] = row
    assert not bad, bad
    assert len(res) >= 36
