"""Static audit of the compiled fused mask kernel (runs without a GPU): no spills, and no instruction
touches an inline-asm-loaded operand register while the load is in flight (tools/audit_asm.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_fused_kernel_asm_audit(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm.py"), "--keep", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failing" in r.stdout
