"""GPU tier: the C++ host mirror (go-tfhe_amd/host/tfhe_gpu.hpp: tfhe::gates / tfhe::evaluator /
tfhe::cloudkey over the C ABI) passes the reference's gate tests and matches the oracle."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_cpp_host_mirror(built):
    import __graft_entry__ as g
    exe = g.build_cpp_host_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert "all checks passed" in r.stdout
