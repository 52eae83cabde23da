"""Build the gfx950 shared library (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.environ.get("TFHE_HIP_LIB") or os.path.join(LIB_DIR, "libtfhe_hip.so")  # env override: kernel experiments
# (source, extra flags): machine-scheduler options per kernel family, each measured (csrc/blind_rotate.hip has the table)
MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
SOURCES = [("tfhe_hip.hip", []), ("blind_rotate.hip", MAX_ILP), ("blind_rotate_oct.hip", MAX_ILP + ["-mllvm", "-enable-post-misched=0"]),
           ("blind_rotate_n2048.hip", [])]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "tfhe_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in [x for x, _ in SOURCES] + HEADERS)


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into lib/libtfhe_hip.so (no-op when up to date)."""
    if os.environ.get("TFHE_HIP_LIB") and os.path.exists(LIB_PATH):
        # an experiment's own binary (tools/build_variant*.sh): never rebuilt from the current sources behind its back
        print(f"[build] using TFHE_HIP_LIB={LIB_PATH} as it is", file=sys.stderr)
        return LIB_PATH
    if not force and not _stale():
        print(f"[build] {os.path.relpath(LIB_PATH)}: up to date with csrc/ (reused)", file=sys.stderr)
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # Several ranks of one node may import the package at once (torch.distributed.run): one of them builds, the others wait
    # on the lock and then find the library up to date.  Objects and library are written under private names and renamed.
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            print(f"[build] {os.path.relpath(LIB_PATH)}: built by another process meanwhile (reused)", file=sys.stderr)
            return LIB_PATH
        print(f"[build] compiling {', '.join(x for x, _ in SOURCES)} for gfx950 -> {os.path.relpath(LIB_PATH)}", file=sys.stderr, flush=True)
        tag = f".{os.getpid()}.tmp"
        objs = []
        procs = []
        for src, extra in SOURCES:
            obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
            cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj + tag]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        for cmd, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        for obj in objs:
            os.replace(obj + tag, obj)
        link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + tag]
        if verbose:
            print(" ".join(link))
        subprocess.run(link, check=True)
        os.replace(LIB_PATH + tag, LIB_PATH)
    return LIB_PATH
