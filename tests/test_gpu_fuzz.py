"""A fixed-seed slice of tools/fuzz_parity.py in the GPU tier: random models / agent counts / horizons / strategies / warm starts /
iteration limits, every result bit for bit against the CPU oracle.  (Round 2 ran 800 random cases of it without a mismatch.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_random_configurations_match_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "11"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "60 cases, seed 11: 0 mismatches" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
