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


def test_fuzz_parity_short_with_the_multi_device_sequence(built):
    """The same run with the test hooks on: about a third of the cases with a pick also go through the WHOLE multi-device sequence --
    ksched_comm_create_local over 2 .. 4 evaluators on the one GPU (TEST-ONLY librccl stand-in), ksched_eval_begin on every replica,
    ksched_gather_buffer, ksched_allgather_bindings_local, ksched_eval_end(gathered_0) -- and must merge to the oracle's bindings and masks."""
    fake = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["make", "-C", ROOT, "-s", "host"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "12", "777001"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, KSCHED_TEST_HOOKS="1", KSCHED_LIB=os.path.join(ROOT, "tests", "cpp", "hooks", "libksched_hip.so"), KSCHED_RCCL_LIB=fake))
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "gathered-over-" in r.stdout and " 0 failures" in r.stdout
