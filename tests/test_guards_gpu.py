"""Guarded device allocations (csrc/dev_guard.h, BSMI_GUARD_MB): forward passes in the three precisions and training steps in both
modes with 4 MiB of 0xFF on both sides of every buffer of the engine -- no zone is written to, no result is poisoned."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_no_kernel_writes_or_reads_past_its_buffers():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BSMI_GUARD_MB="4")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "guard_worker.py")], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "guards intact" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
