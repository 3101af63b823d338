"""The best-fit pick on the pods that are hard for it -- the AND of their rows is sparse: several selective label keys, a value no
node carries (decided in the first stage without a scan), a cpu request only a handful of nodes can hold, no feasible node at all
(the scan runs to the end of the snapshot) -- at snapshot sizes where the scan spans several wave rounds, one and two stages, every
hand-over point of the first stage; all against the oracle."""
import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import FIT, PICK_BESTFIT, SEL, SEL_NEVER, TAINT, _lib, synth
from oracle import capi

pytestmark = pytest.mark.gpu


def want_of(c, sel, req_cpu, req_mem, flags):
    return capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, c.node_taints, req_cpu, req_mem, sel, c.pod_tol, None, flags | PICK_BESTFIT)[2]


@pytest.mark.parametrize("N", [700, 5_000, 40_000, 70_001])
def test_sparse_and_infeasible_pods_one_and_two_stages(evaluator, N):
    ev = evaluator
    P = 3000
    c = synth.make_cluster(P, N, n_keys=8, n_taints=16, seed=1000 + N)
    rng = np.random.default_rng(N)
    sel = c.pod_sel.copy()
    req_cpu, req_mem = c.req_cpu.copy(), c.req_mem.copy()
    card = c.node_labels.max(axis=1)
    # sparse ANDs: every third pod constrains the three most selective keys with values nodes do carry
    for k in (5, 6, 7):
        sel[k, ::3] = rng.integers(1, int(card[k]) + 1, size=sel[k, ::3].shape)
    sel[7, 1::50] = SEL_NEVER                      # a value no node carries
    sel[6, 2::50] = np.uint32(int(card[6]) + 7)    # an id beyond the key's largest
    req_cpu[3::11] = np.sort(c.avail_cpu)[-3]       # only the few largest nodes can hold these
    req_cpu[4::97] = c.avail_cpu.max() + 1         # nothing can
    req_mem[5::89] = c.avail_mem.max() + 1
    ev.set_nodes(**c.node_columns())
    try:
        for flags in (FIT | SEL | TAINT, FIT | SEL, SEL | TAINT, FIT):
            want = want_of(c, sel, req_cpu, req_mem, flags)
            for stages in (2, 1):
                ev.set_option(_lib.OPT_BESTFIT_STAGES, stages)
                # bits 12-15: the first stage hands over after this many 64-byte blocks of candidate words (default 2); bit 11: unused since round 3
                for dbg in ((0, 1 << 12, 2 << 12, 3 << 12, 5 << 12, 15 << 12) if stages == 2 else (0,)):
                    ev.set_option(_lib.OPT_DEBUG, dbg)
                    r = ev.eval(req_cpu, req_mem, sel, c.pod_tol, None, flags | PICK_BESTFIT, want_mask=False)
                    assert np.array_equal(r.binding, want), (N, flags, stages, hex(dbg), int((r.binding != want).sum()))
        assert (want_of(c, sel, req_cpu, req_mem, FIT | SEL | TAINT) == -1).sum() > P // 100  # the case this test is about exists
    finally:
        ev.set_option(_lib.OPT_DEBUG, 0)
        ev.set_option(_lib.OPT_BESTFIT_STAGES, 0)


@pytest.mark.parametrize("what", ["nothing_fits", "only_the_largest_nodes"])
def test_every_pod_handed_over(evaluator, what):
    """Every pod of the batch is still undecided after the first stage (its cpu request excludes all, or all but the few largest, nodes): the
    hand-over sub-lists fill to their capacity and the second stage's grid (a quarter of it) walks them; all against the oracle."""
    ev = evaluator
    P, N = 20_000, 20_000
    c = synth.make_cluster(P, N, n_keys=8, n_taints=16, seed=4242)
    req_cpu = np.full(P, c.avail_cpu.max() + (1 if what == "nothing_fits" else 0), dtype=np.int64)
    req_mem = np.minimum(c.req_mem, np.sort(c.avail_mem)[N // 4])  # (so that `start` is early and the scan is long)
    sel = np.zeros_like(c.pod_sel)
    ev.set_nodes(**c.node_columns())
    try:
        ev.set_option(_lib.OPT_BESTFIT_STAGES, 2)
        for flags in (FIT | TAINT, FIT):
            want = want_of(c, sel, req_cpu, req_mem, flags)
            for _ in range(4):  # (the counter sets rotate over three slots)
                r = ev.eval(req_cpu, req_mem, sel, c.pod_tol, None, flags | PICK_BESTFIT, want_mask=False)
                assert np.array_equal(r.binding, want), (what, flags, int((r.binding != want).sum()))
            assert ((want == -1).all() if what == "nothing_fits" else (want >= 0).any())
    finally:
        ev.set_option(_lib.OPT_BESTFIT_STAGES, 0)


def test_two_stage_pick_after_snapshot_updates_many_calls(evaluator):
    """ksched_update_nodes marks the best-fit structures stale; the hand-over counters (128 sub-lists) rotate over three sets (each call zeroes
    the next call's set instead of a memset launch): many consecutive two-stage calls stay == oracle."""
    ev = evaluator
    c = synth.make_cluster(2000, 9000, n_keys=8, n_taints=16, seed=77)
    rng = np.random.default_rng(5)
    cpu, mem = c.avail_cpu.copy(), c.avail_mem.copy()
    ev.set_nodes(cpu, mem, c.node_labels, c.node_taints)
    ev.set_option(_lib.OPT_BESTFIT_STAGES, 2)
    try:
        for step in range(7):
            want = capi.eval_encoded(cpu, mem, c.node_labels, c.node_taints, c.req_cpu, c.req_mem, c.pod_sel, c.pod_tol, None, FIT | SEL | TAINT | PICK_BESTFIT)[2]
            r = ev.eval(c.req_cpu, c.req_mem, c.pod_sel, c.pod_tol, None, FIT | SEL | TAINT | PICK_BESTFIT, want_mask=False)
            assert np.array_equal(r.binding, want), step
            idx = rng.choice(c.N, size=400, replace=False).astype(np.uint32)
            cpu[idx] = rng.integers(0, 64_000, size=idx.size)
            mem[idx] = rng.integers(0, 1 << 37, size=idx.size)
            ev.update_nodes(idx, cpu[idx], mem[idx])
    finally:
        ev.set_option(_lib.OPT_BESTFIT_STAGES, 0)
