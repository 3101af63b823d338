"""The fit part of the per-tile bitmap index (csrc/tile_index.hpp) checked on the host, no GPU: a scalar emulation of
what the fused kernel does with it (rank search, cnt[rank], one row chunk per sub-tile) must equal `req <= avail`
(src/predicates.rs:42) for every (request, node) pair -- tests/cpp/index_tests.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "index_tests")


def test_index_fit_tables_on_host():
    if os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        subprocess.check_call(["make", "-C", ROOT, "-s", "tests/cpp/index_tests"])
    assert os.path.exists(BIN), "tests/cpp/index_tests has not been built (make host)"
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed check(s)" in r.stdout
