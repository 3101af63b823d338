"""ksched_mask_alloc / ksched_mask_free / ksched_mask_probe_report (ABI 6): mask buffers placed by the library.

The reference never materialises a mask (check_node_validity answers one pair at a time, src/predicates.rs:63-77); the batched path's mask is its own
artefact, and where it lies decides the rate of the kernel that writes it (profiles/r06_mask_alloc.md).  Checked here: every allocation path hands out
a buffer the evaluator writes the SAME words into (== the oracle), the pitch it reports is the one ksched_eval_device_pitched expects, the probe-and-keep
path keeps exactly one candidate and reports what it measured, and the error behaviour of the two entry points.
"""
import ctypes as C
import os

import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import FIT, PICK_SAMPLED, SEL, Evaluator, KschedError, _lib, synth
from oracle import capi

pytestmark = pytest.mark.gpu


def _case(P=3000, N=2500, seed=0x6A):
    c = synth.make_cluster(P, N, n_keys=8, n_taints=0, seed=seed)
    pc = c.pod_columns()
    return c, pc


def _oracle(c, pc, flags):
    return capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], flags)


PRODUCT_PATHS = [_lib.MASK_ALLOC_AUTO, _lib.MASK_ALLOC_PLAIN, _lib.MASK_ALLOC_PROBE]
MEASUREMENT_PATHS = [_lib.MASK_ALLOC_VMM, _lib.MASK_ALLOC_VMM_MIN, _lib.MASK_ALLOC_CONTIGUOUS, _lib.MASK_ALLOC_SCATTER_2M, _lib.MASK_ALLOC_SCATTER_16M]


def _same_mask_through(how):
    import torch
    c, pc = _case()
    flags = FIT | SEL | PICK_SAMPLED
    o_feas, _, o_bind = _oracle(c, pc, flags)
    with Evaluator(0) as ev:
        ev.set_nodes(**c.node_columns())
        dev = torch.device("cuda", 0)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
        d = (t(pc["req_cpu_milli"], np.int64), t(pc["req_mem_bytes"], np.int64), t(pc["sel_val_ids"], np.int32), None, t(pc["samples"], np.int32))
        mask = ev.alloc_mask(c.P, how=how)
        assert tuple(mask.shape) == (c.P, ev.W) and mask.stride(0) == ev._lib.ksched_mask_pitch(ev.n) and mask.stride(1) == 1
        assert mask.data_ptr() % 128 == 0, "rows start on cache-line boundaries"
        bind = torch.full((c.P,), -9, dtype=torch.int32, device=dev)
        for kernel in ("fused", "direct"):
            ev.set_kernel(kernel)
            mask.fill_(-1)
            ev.eval_device(*d, flags, out_feasible=mask, out_binding=bind)
            torch.cuda.synchronize()
            assert np.array_equal(mask.contiguous().cpu().numpy().view(np.uint64), o_feas), (kernel, _lib.MASK_ALLOC_NAMES[how])
            assert np.array_equal(bind.cpu().numpy(), o_bind)
        rep = ev.mask_probe_report()
        if how == _lib.MASK_ALLOC_PROBE:
            assert 2 <= rep.size <= 6 and (rep > 0).all(), "the probe times every candidate and says what it read"
        elif how != _lib.MASK_ALLOC_AUTO:
            assert rep.size == 0
        del mask  # (ksched_mask_free through the tensor's owner; the evaluator is still open)


@pytest.mark.parametrize("how", PRODUCT_PATHS, ids=lambda h: _lib.MASK_ALLOC_NAMES[h])
def test_the_product_paths_hold_the_same_mask(built, how):
    """AUTO, PLAIN and PROBE (hipMalloc underneath, all three): what every mask of the GPU suite and of bench.py comes from."""
    _same_mask_through(how)


TEST_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "hooks", "libksched_hip.so")


@pytest.mark.parametrize("how", MEASUREMENT_PATHS, ids=lambda h: _lib.MASK_ALLOC_NAMES[h])
def test_a_measurement_path_holds_the_same_mask_in_the_test_build(built, how):
    """The paths tools/alloc_probe.py measures (HIP's virtual-memory API, one contiguous physical range, scattered pieces) exist in the TEST build of
    the library only (tests/cpp/test_hooks.cpp): same words, same bindings -- each in a fresh process that loads that build ($KSCHED_LIB), also because
    on ROCm 7.0 a virtual-memory mapping made right after a contiguous allocation was freed in the same process returned STALE shader reads on its
    first use (the data in memory -- read back by the DMA engine -- was right; profiles/r06_mask_alloc.md section 3).  The shipped library answers
    KSCHED_E_UNSUPPORTED for all of them."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"import tests.test_gpu_mask_alloc as t; t._same_mask_through({how}); print('path ok')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, KSCHED_LIB=TEST_LIB))
    assert r.returncode == 0 and "path ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    with Evaluator(0) as ev:  # the shipped library (this process)
        c, _ = _case(P=8, N=130)
        ev.set_nodes(**c.node_columns())
        with pytest.raises(KschedError) as ei:
            ev.alloc_mask(10, how=how)
        assert ei.value.code == _lib.E_UNSUPPORTED and "test build" in str(ei.value)


def test_the_probe_keeps_one_buffer_and_frees_the_rest(built):
    """Six candidates of a C4-shard-sized mask (153 MiB) are alive only while the probe runs: what the device holds afterwards is one mask more than
    before, and AUTO is the probe from 128 MiB up and a plain allocation below that."""
    import torch
    c = synth.make_config("C4", P=125_000, N=10_000)
    with Evaluator(0) as ev:
        ev.set_nodes(**c.node_columns())
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        big = ev.alloc_mask(125_000)  # AUTO: 125 000 rows x 160 words x 8 B = 153 MiB -> probe-and-keep
        rep = ev.mask_probe_report()
        assert rep.size == 6 and (rep > 5.0).all() and rep.max() < 20 * rep.min(), rep
        free1, _ = torch.cuda.mem_get_info()
        held = free0 - free1
        assert 150 << 20 <= held <= 260 << 20, f"one mask (plus what the probe's scratch keeps), not six: {held / 2**20:.0f} MiB"
        ev.set_option(_lib.OPT_MASK_PROBE, 3)
        other = ev.alloc_mask(125_000)
        assert ev.mask_probe_report().size == 3
        small = ev.alloc_mask(10_000)  # 12 MiB: AUTO = plain
        assert ev.mask_probe_report().size == 0
        ev.set_option(_lib.OPT_MASK_PROBE, 1)  # probing off
        plain = ev.alloc_mask(125_000)
        assert ev.mask_probe_report().size == 0
        with pytest.raises(KschedError):
            ev.set_option(_lib.OPT_MASK_PROBE, 0)
        with pytest.raises(KschedError):
            ev.set_option(_lib.OPT_MASK_PROBE, 17)
        del big, other, small, plain
        torch.cuda.synchronize()
        free2, _ = torch.cuda.mem_get_info()
        assert free0 - free2 <= 64 << 20, "everything handed back (ksched_mask_free)"


def test_error_behaviour_of_the_two_entry_points(built):
    lib = _lib.load()
    with Evaluator(0) as ev:
        ptr, pitch = C.c_void_p(), C.c_uint32(7)
        # before a snapshot: the pitch follows the node count
        assert lib.ksched_mask_alloc(ev._h, 10, _lib.MASK_ALLOC_AUTO, C.byref(ptr), C.byref(pitch)) == _lib.E_STATE and not ptr.value
        assert b"ksched_set_nodes" in lib.ksched_last_error(ev._h)
        c, _ = _case(P=8, N=130)
        ev.set_nodes(**c.node_columns())
        assert lib.ksched_mask_alloc(ev._h, 10, _lib.MASK_ALLOC_LAST + 1, C.byref(ptr), C.byref(pitch)) == _lib.E_INVAL
        for removed in (3, 6, 7, 10):  # paths measured in round 6 and removed (a memory pool among them: profiles/r06_mask_alloc.md section 3)
            assert lib.ksched_mask_alloc(ev._h, 10, removed, C.byref(ptr), C.byref(pitch)) == _lib.E_INVAL and not ptr.value
        assert lib.ksched_mask_alloc(ev._h, 10, _lib.MASK_ALLOC_PLAIN, None, None) == _lib.E_INVAL
        assert lib.ksched_mask_alloc(None, 10, _lib.MASK_ALLOC_PLAIN, C.byref(ptr), None) == _lib.E_INVAL
        # p = 0 is a valid (empty) mask; the pitch pointer is optional
        assert lib.ksched_mask_alloc(ev._h, 0, _lib.MASK_ALLOC_PLAIN, C.byref(ptr), None) == _lib.OK and ptr.value
        assert lib.ksched_mask_free(ev._h, ptr) == _lib.OK
        assert lib.ksched_mask_free(ev._h, ptr) == _lib.E_INVAL, "a pointer is freed once"
        assert b"ksched_mask_free" in lib.ksched_last_error(ev._h)
        assert lib.ksched_mask_free(ev._h, None) == _lib.OK  # like free(NULL)
        assert lib.ksched_mask_free(ev._h, C.c_void_p(0x1000)) == _lib.E_INVAL  # not ours
        assert lib.ksched_mask_alloc(ev._h, 10, _lib.MASK_ALLOC_VMM, C.byref(ptr), C.byref(pitch)) == _lib.E_UNSUPPORTED and not ptr.value  # (a measurement path: test build only)
        assert lib.ksched_mask_alloc(ev._h, 10, _lib.MASK_ALLOC_PLAIN, C.byref(ptr), C.byref(pitch)) == _lib.OK
        assert pitch.value == lib.ksched_mask_pitch(130) == 16
        # left to ksched_destroy (the `with` block): no leak report, no crash
    with Evaluator(0) as ev2:  # a second context does not see the first one's buffers
        c, _ = _case(P=8, N=130)
        ev2.set_nodes(**c.node_columns())
        assert lib.ksched_mask_free(ev2._h, ptr) == _lib.E_INVAL
