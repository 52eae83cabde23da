"""Build the gfx950 shared library (hipcc cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.environ.get("TFHE_HIP_LIB") or os.path.join(LIB_DIR, "libtfhe_hip.so")  # env override: kernel experiments
SOURCES = ["tfhe_hip.hip"]
HEADERS = ["kernels.hpp", "kernels_n2048.hpp", "negacyclic_fft.hpp", os.path.join("..", "..", "include", "tfhe_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into lib/libtfhe_hip.so (no-op when up to date)."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH
