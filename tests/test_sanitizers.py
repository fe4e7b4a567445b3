"""AddressSanitizer + UBSan over the host C / C++ of the path (missing 5): `make asan` in
bootstrapper_amd/csrc (chunk_codec.cpp, agglo_host.cpp, flood_host.cpp -- without the device half of libbsmi; GPU ASan is not
available on the MI355X pool) and in oracle/ (seg_ref.c), exercised by tests/sanitizer_worker.py in a child process that has
libasan preloaded.  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_code_and_oracle_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    subprocess.run(["make", "-C", os.path.join(ROOT, "bootstrapper_amd", "csrc"), "asan"], check=True, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=asan, BSMI_ORACLE_SO=os.path.join(ROOT, "oracle", "_build", "libsegref_asan.so"),
               # CPython's own allocations at exit are not ours to account for; everything else is fatal
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZERS OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
