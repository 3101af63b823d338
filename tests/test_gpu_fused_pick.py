"""The sampled pick riding in the fused mask launch (KSCHED_OPT_FUSED_PICK, kernels_fused.hpp "PICK") and the
no-unwind rule of the C ABI (KSCHED_OPT_FAULT).

select_node_for_pod (src/main.rs:51-71): ATTEMPTS draws with replacement, the first feasible one wins, none -> NoNodeFound.
Three implementations must agree on every pod: the pick inside the mask launch, the stand-alone launch (k_select_sampled)
and the oracle's scalar loop -- and the mask written by the same launch must not notice that the pick rode along.
"""
import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import FIT, PICK_SAMPLED, SEL, TAINT, WANT_FIT_MASK, KschedError, _lib, synth
from oracle import capi

pytestmark = pytest.mark.gpu


def oracle_eval(c, flags):
    return capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels if c.n_keys else None, c.node_taints if c.n_taints else None,
                             c.req_cpu, c.req_mem, c.pod_sel if c.n_keys else None, c.pod_tol if c.n_taints else None, c.samples, flags)


def run(ev, c, flags, ride):
    ev.set_option(_lib.OPT_FUSED_PICK, 1 if ride else 0)
    try:
        pc = c.pod_columns()
        r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], pc["tolerations"], pc["samples"], flags)
        return r, ev.last_pick, ev.last_kernel
    finally:
        ev.set_option(_lib.OPT_FUSED_PICK, 1)


@pytest.mark.parametrize("P,N,attempts", [(1, 1, 5), (63, 65, 5), (300, 200, 5), (1000, 4097, 5), (5000, 1000, 1), (4099, 2500, 8),
                                          (2500, 700, 11), (20_000, 5_000, 5), (70_000, 1_100, 5)])
@pytest.mark.parametrize("flags", [FIT, FIT | SEL, FIT | SEL | TAINT, SEL, 0])
def test_riding_pick_equals_standalone_pick_equals_oracle(evaluator, P, N, attempts, flags):
    ev = evaluator
    ev.set_kernel("fused")
    try:
        c = synth.make_cluster(P, N, n_keys=8, n_taints=16, seed=P * 31 + N * 7 + attempts, attempts=attempts)
        ev.set_nodes(**c.node_columns())
        f = flags | PICK_SAMPLED
        a, how_a, kern_a = run(ev, c, f, ride=True)
        b, how_b, _ = run(ev, c, f, ride=False)
        feas, _, bind = oracle_eval(c, f)
        assert kern_a == "fused" and how_a == "fused" and how_b == "select"
        assert np.array_equal(a.binding, bind), "riding pick vs oracle"
        assert np.array_equal(b.binding, bind), "stand-alone pick vs oracle"
        assert np.array_equal(a.feasible, feas) and np.array_equal(b.feasible, feas), "the mask does not notice the pick"
    finally:
        ev.set_kernel("auto")


def test_riding_pick_out_of_range_draws_and_no_feasible_node(evaluator):
    """Draws >= N are infeasible draws (include/ksched.h); a pod no draw fits gets -1 (NoNodeFound, src/main.rs:117)."""
    ev = evaluator
    c = synth.make_cluster(3000, 900, n_keys=8, n_taints=0, seed=4242)
    rng = np.random.default_rng(7)
    c.samples[rng.random(c.samples.shape) < 0.3] = np.uint32(900 + 5)   # out of range
    c.samples[:50, :] = np.uint32(0xFFFFFFFF)                          # a pod whose every draw is out of range
    c.req_cpu[100:150] = np.int64(1) << 60                             # pods nothing can hold
    ev.set_nodes(**c.node_columns())
    ev.set_kernel("fused")
    try:
        a, how, _ = run(ev, c, FIT | SEL | PICK_SAMPLED, ride=True)
        _, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
        assert how == "fused"
        assert np.array_equal(a.binding, bind)
        assert (a.binding[:50] == -1).all() and (a.binding[100:150] == -1).all()
    finally:
        ev.set_kernel("auto")


def test_the_pick_does_not_ride_where_it_cannot(evaluator):
    """A second mask (WANT_FIT_MASK), the direct kernel, a bindings-only request: the stand-alone launch, same bindings."""
    ev = evaluator
    c = synth.make_cluster(2000, 1500, n_keys=8, n_taints=0, seed=99)
    ev.set_nodes(**c.node_columns())
    pc = c.pod_columns()
    _, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED | WANT_FIT_MASK)
    assert ev.last_pick == "select" and np.array_equal(r.binding, bind)
    r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED, want_mask=False)
    assert ev.last_pick == "select" and np.array_equal(r.binding, bind)
    ev.set_kernel("direct")
    try:
        r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED)
        assert ev.last_pick == "select" and ev.last_kernel == "direct" and np.array_equal(r.binding, bind)
    finally:
        ev.set_kernel("auto")
    r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED)
    assert ev.last_pick == "fused" and np.array_equal(r.binding, bind)


def test_riding_pick_on_device_buffers_repeated_steps(evaluator):
    """The bench's form: device-resident inputs, pitched mask rows, the same launch over and over on one stream."""
    import torch
    ev = evaluator
    c = synth.make_config("C3", P=30_000)
    ev.set_nodes(**c.node_columns())
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel, d_smp = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.samples, np.int32)
    mask = ev.alloc_mask(c.P, pitched=True)
    out = torch.full((c.P,), -7, dtype=torch.int32, device=dev)
    feas, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    for _ in range(5):
        out.fill_(-7)
        ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, FIT | SEL | PICK_SAMPLED, out_feasible=mask, out_binding=out)
        torch.cuda.synchronize()
        assert ev.last_pick == "fused"
        assert np.array_equal(out.cpu().numpy(), bind)
    assert np.array_equal(mask.cpu().numpy().view(np.uint64), feas)


# ---- nothing unwinds across the C ABI (include/ksched.h "Conventions") -------------------------------------

@pytest.mark.parametrize("kind,code", [(1, _lib.E_NOMEM), (2, _lib.E_INVAL)])
def test_an_exception_inside_the_library_comes_back_as_a_code(evaluator, kind, code):
    """KSCHED_OPT_FAULT makes the next entry into the library's C++ throw (std::bad_alloc / std::runtime_error): the call returns
    KSCHED_E_NOMEM / KSCHED_E_INVAL with ksched_last_error set -- this process is still alive to assert it -- and the ctx works on."""
    ev = evaluator
    c = synth.make_cluster(500, 300, n_keys=8, n_taints=0, seed=5)
    ev.set_nodes(**c.node_columns())
    pc = c.pod_columns()
    args = (pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED)
    good = ev.eval(*args)
    # (a) inside an evaluation
    ev.set_option(_lib.OPT_FAULT, kind)
    with pytest.raises(KschedError) as ei:
        ev.eval(*args)
    assert ei.value.code == code and "exception inside the library" in str(ei.value)
    again = ev.eval(*args)  # one shot: the next call is normal, on an intact snapshot
    assert np.array_equal(again.feasible, good.feasible) and np.array_equal(again.binding, good.binding)
    # (b) inside ksched_update_nodes, before anything changed: the snapshot stays valid and unchanged
    ev.set_option(_lib.OPT_FAULT, kind)
    with pytest.raises(KschedError) as ei:
        ev.update_nodes(np.array([1, 2], dtype=np.uint32), np.array([0, 0], dtype=np.int64), np.array([0, 0], dtype=np.int64))
    assert ei.value.code == code
    again = ev.eval(*args)
    assert np.array_equal(again.feasible, good.feasible)
    # (c) inside ksched_set_nodes: a half-built snapshot is never evaluated (KSCHED_E_STATE) until the next successful set_nodes
    ev.set_option(_lib.OPT_FAULT, kind)
    with pytest.raises(KschedError) as ei:
        ev.set_nodes(**c.node_columns())
    assert ei.value.code == code
    with pytest.raises(KschedError) as ei:
        ev.eval(*args)
    assert ei.value.code == _lib.E_STATE
    ev.set_nodes(**c.node_columns())
    again = ev.eval(*args)
    assert np.array_equal(again.feasible, good.feasible) and np.array_equal(again.binding, good.binding)
    # (d) a deeper fault point: skip one (the evaluation's own), throw at the next (the following call)
    ev.set_option(_lib.OPT_FAULT, kind | (1 << 8))
    ev.eval(*args)
    with pytest.raises(KschedError):
        ev.eval(*args)
    ev.eval(*args)
