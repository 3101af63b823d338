"""A short run of the randomised differential test (tools/fuzz_parity.py): random shapes, key counts / cardinalities (rows, list
keys, more than eight keys), taints, predicate subsets, both picks, snapshot updates between evaluations, both kernels -- every
mask word and binding against the oracle.  (A 240 s run of the same tool: 1217 cases, 0 failures.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_parity_short(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "12", "20260923"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "0 failures" in r.stdout
