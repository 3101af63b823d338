"""The C-ABI shared library loads and exports every symbol include/ksched.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ksched.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ksched_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("ksched_create", "ksched_destroy", "ksched_set_nodes", "ksched_eval", "ksched_eval_device", "ksched_strerror"):
        assert must in names


def test_library_exports_every_declared_symbol(built):
    from kube_scheduler_rs_reference_amd import _lib
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in ksched.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.SYMBOLS) == declared_functions()


def test_constants_match_header(built):
    from kube_scheduler_rs_reference_amd import _lib
    text = open(HEADER).read()
    defs = {m.group(1): m.group(2) for m in re.finditer(r"#define\s+KSCHED_([A-Z_0-9]+)\s+\(?(-?(?:0x)?[0-9A-Fa-f]+)u?\)?", text)}
    for k, v in defs.items():
        if hasattr(_lib, k):
            assert getattr(_lib, k) == int(v, 0), k
    assert _lib.load().ksched_abi_version() == int(defs["ABI_VERSION"], 0)
    assert _lib.load().ksched_mask_words(0) == 0
    assert _lib.load().ksched_mask_words(64) == 1
    assert _lib.load().ksched_mask_words(65) == 2


def test_no_cpu_fallback(built):
    """Without a GPU the product refuses to run instead of falling back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kube_scheduler_rs_reference_amd import Evaluator, KschedError, _lib
    with pytest.raises(KschedError) as ei:
        Evaluator(0)
    assert ei.value.code == _lib.E_NODEVICE


def test_null_arguments_are_errors_not_crashes(built):
    from kube_scheduler_rs_reference_amd import _lib
    lib = _lib.load()
    assert lib.ksched_create(None, 0) == _lib.E_INVAL
    assert lib.ksched_set_nodes(None, 0, None, None, None, 0, None) == _lib.E_INVAL
    assert lib.ksched_update_nodes(None, 0, None, None, None) == _lib.E_INVAL
    assert lib.ksched_eval(None, 0, None, None, None, None, None, 0, 0, None, None, None) == _lib.E_INVAL
    lib.ksched_destroy(None)
    assert b"no CPU fallback" in lib.ksched_strerror(_lib.E_NODEVICE)


def test_shard_bounds_is_the_one_definition_of_the_row_split(built):
    """ksched_shard_bounds (C ABI) == dist.shard_bounds (Python) for every (P, world, rank): the C++ host mirror, the Rust overlay and the
    one-process-per-GPU scheduler all cut a batch's pod rows the same way (contiguous rows, ceil(P / n) per device)."""
    from kube_scheduler_rs_reference_amd import _lib
    from kube_scheduler_rs_reference_amd.dist import shard_bounds
    lib = _lib.load()
    lo, hi, cpr = C.c_uint32(), C.c_uint32(), C.c_uint32()
    for world in range(1, 10):
        for P in list(range(0, 70)) + [999, 1000, 1001, 100_000, 1_000_000, (1 << 32) - 1]:
            covered = 0
            for rank in range(world):
                lib.ksched_shard_bounds(P, world, rank, C.byref(lo), C.byref(hi), C.byref(cpr))
                assert (lo.value, hi.value, cpr.value) == shard_bounds(P, world, rank), (P, world, rank)
                assert lo.value == covered and hi.value - lo.value <= cpr.value
                covered = hi.value
            assert covered == P
    lib.ksched_shard_bounds(10, 4, 1, None, None, None)  # null outputs are allowed
    assert lib.ksched_device_count() >= 0
    assert lib.ksched_eval_begin(None, 0, None, None, None, 0, None, None, 0, 0, None, None, 0, None, None) == _lib.E_INVAL
    assert lib.ksched_gather_buffer(None, 0, None) == _lib.E_INVAL and lib.ksched_eval_end(None, None, 0, None) == _lib.E_INVAL


def test_reason_helper(built):
    import numpy as np
    from kube_scheduler_rs_reference_amd import _lib
    lib = _lib.load()
    feas = np.array([0b0001], dtype=np.uint64)
    fit = np.array([0b0011], dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    fl = _lib.FIT | _lib.SEL
    assert lib.ksched_reason(p(feas), p(fit), 0, fl) == _lib.REASON_OK
    assert lib.ksched_reason(p(feas), p(fit), 1, fl) == _lib.REASON_NODE_SELECTOR_MISMATCH
    assert lib.ksched_reason(p(feas), p(fit), 2, fl) == _lib.REASON_NOT_ENOUGH_RESOURCES  # fit first, src/predicates.rs:68-70


def test_product_does_not_import_the_oracle():
    """The product package must never route through oracle/ (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "kube_scheduler_rs_reference_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".rs")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle.h" not in src and "liboracle" not in src, f


def test_the_shipped_library_carries_no_test_hooks(built):
    """The hooks the tests use (an RCCL stand-in named by $KSCHED_RCCL_LIB, fault injection, k replicas of one device) live in tests/cpp/test_hooks.cpp,
    which is linked into tests/cpp/hooks/libksched_hip.so ONLY: the shipped library neither defines the hook functions nor contains the variable's
    name, so no environment can redirect it."""
    import subprocess
    shipped = os.path.join(ROOT, "kube_scheduler_rs_reference_amd", "libksched_hip.so")
    test_build = os.path.join(ROOT, "tests", "cpp", "hooks", "libksched_hip.so")
    assert os.path.exists(shipped) and os.path.exists(test_build)

    def defined(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    ds, dt = defined(shipped), defined(test_build)
    hooks = {"ksched_test_hooks_enabled", "ksched_test_hooks_linked", "ksched_test_rccl_lib", "ksched_test_mask_alloc", "ksched_test_mask_release"}
    assert not (hooks & ds), hooks & ds
    assert hooks <= dt
    assert {s_ for s_ in ds if s_.startswith("ksched_")} == {s_ for s_ in dt if s_.startswith("ksched_")} - hooks  # otherwise the same library
    text = open(shipped, "rb").read()
    assert b"KSCHED_RCCL_LIB" not in text and b"KSCHED_TEST_HOOKS" not in text
    assert b"KSCHED_RCCL_LIB" in open(test_build, "rb").read()
    # ... nor HIP's virtual-memory API (the measurement paths of ksched_mask_alloc: test build only)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", shipped], capture_output=True, text=True, check=True).stdout
    assert "hipMemCreate" not in undefined and "hipMemMap" not in undefined and "hipMemAddressReserve" not in undefined
    assert "hipMemCreate" in subprocess.run(["nm", "-D", "--undefined-only", test_build], capture_output=True, text=True, check=True).stdout
