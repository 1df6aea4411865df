"""The CPU pieces of the host that parse files (MAT5, PNG) and pre-process depth, built with AddressSanitizer + UBSan and run on
well-formed, truncated and corrupted inputs (srmeetsps-cuda_amd/host/SelfTest.cpp).  The GPU pool refuses sanitizer runs; this is
where they can run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "srmeetsps-cuda_amd", "host")


def test_host_file_parsers_under_asan_and_ubsan(tmp_path):
    b = subprocess.run(["make", "-C", HOST, "sanitize"], capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "ubsan" in b.stderr.lower() or "sanitize" in b.stderr.lower()):
        pytest.skip("this toolchain has no sanitizer runtime: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(HOST, "selftest_sanitized"), str(tmp_path)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "SelfTest ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
