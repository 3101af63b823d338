"""Host-side sanitizer runs of the C++ mirror (make sanitize -> build/san/*; tools/sanitize.sh): AddressSanitizer + UBSan + the leak check, and
ThreadSanitizer, over the mirror's own tests.  Here: the CPU half (wire format, quantities, encoder, snapshot builder incl. the staged update and
the worker pool, PodBatcher / run_batches with their producer threads).  The device halves -- every `host_tests` mode and a C3-size batch through
`objects_eval` -- run on a GPU box (`bash tools/sanitize.sh <out>`; this round's logs: profiles/r06_sanitizers_*).

The sanitizer builds take two minutes and are not part of `make all`, and ThreadSanitizer's run time depends on what else the machine is doing: the
test is opt-in -- KSCHED_TEST_SANITIZE=1 builds the binaries (when a compiler is there) and runs them; without it the test is skipped."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "build", "san")


def _have():
    return all(os.path.exists(os.path.join(SAN, b)) for b in ("host_tests_asan", "host_tests_tsan"))


def test_host_mirror_cpu_half_under_the_sanitizers(tmp_path):
    if os.environ.get("KSCHED_TEST_SANITIZE") != "1":
        pytest.skip("opt-in: KSCHED_TEST_SANITIZE=1 (builds build/san/* with `make sanitize` and runs the CPU half under ASan + UBSan and TSan)")
    if os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-C", ROOT, "-s", "-j8", "sanitize"])
    if not _have():
        pytest.skip("build/san/host_tests_{asan,tsan} not built (make sanitize)")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize.sh"), str(tmp_path), "cpu"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout + r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("asan cpu", "tsan cpu"))]
    assert len(lines) == 2
    for ln in lines:
        assert "exit 0" in ln and "0 failed check(s)" in ln
        assert "AddressSanitizer errors 0, UBSan reports 0, ThreadSanitizer reports 0" in ln
