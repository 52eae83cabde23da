"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU tier).  SURVEY.md section 5: the reference runs neither a race
detector nor a sanitizer; the restatement every parity claim rests on gets one -- tests/cpp/oracle_sanitize.c walks every entry point of
oracle/tfhe_oracle.h (key generation, encryption, all eleven gates through the threaded batch call, the exact-integer chain, programmable
bootstraps at three ring shapes, the transforms) at reduced LWE dimensions and checks decryptions while the sanitizers watch."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    exe = os.path.join(tempfile.mkdtemp(prefix="orc_san_"), "oracle_sanitize")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-fopenmp", "-std=c11",
           "-D_GNU_SOURCE", os.path.join(ROOT, "oracle", "tfhe_oracle.c"), os.path.join(ROOT, "oracle", "tfhe_harness.c"),
           os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.c"), "-lm", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "ubsan" in b.stderr.lower()) and "cannot find" in b.stderr:
        pytest.skip("this gcc has no sanitizer runtimes: " + b.stderr.strip().splitlines()[-1])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "oracle under ASan + UBSan: ok" in r.stdout
