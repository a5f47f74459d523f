"""Determinism of the staging protocols (LDS-DMA pieces requested by inline assembly, row barriers, key-switch table
pipeline): the same launch repeated must give bit-identical words every time.  A race would show as a rare mismatch; the
parity tests run every shape once or twice, this runs each a few dozen times (tools/determinism_soak.py, a process of its
own like the other tools so that its parameter-set and N = 2048 initialisations do not leak into other tests)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_repeated_launches_are_bit_identical():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "determinism_soak.py"), "24"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "TOTAL differing launches: 0" in p.stdout
