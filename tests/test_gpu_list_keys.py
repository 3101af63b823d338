"""High-cardinality label keys (the round-1 cliff: one kubernetes.io/hostname-style key dropped the mask kernel 26x to the direct
kernel).  Such keys are now kept per tile as sorted LISTS next to the bitmap rows (csrc/tile_index.hpp): the fused kernel stays
applicable.  Parity vs the oracle, on both kernels, for: one value per node (hostname), a key inside and outside the first eight
columns, two list keys, mid-cardinality keys whose ranges are longer than the unchecked path walks (checked path), SEL_NEVER, more
than eight constrained keys together with a list key, predicate subsets, taints + fit mask (the widest instantiation), ragged tiles;
the device-built lists equal the host specification; and a third such key falls back to the direct kernel with a reason."""
import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import FIT, PICK_BESTFIT, PICK_SAMPLED, SEL, SEL_NEVER, TAINT, WANT_FIT_MASK, _lib, synth
from oracle import capi

pytestmark = pytest.mark.gpu


def with_keys(c, cards, p_constrain=0.15, seed=0, at=None):
    """Add label columns of the given cardinalities to cluster `c` (columns appended, or replacing column `at`)."""
    rng = np.random.default_rng(seed)
    labs, sels = [c.node_labels], [c.pod_sel]
    for card in cards:
        if card >= c.N:  # one value per node (hostname)
            col = (rng.permutation(c.N) + 1).astype(np.uint32)
        else:
            col = rng.integers(0, card + 1, c.N).astype(np.uint32)  # 0 = absent on some nodes
        mx = int(col.max())
        sel = np.where(rng.random(c.P) < p_constrain, rng.integers(1, mx + 3, c.P), 0).astype(np.uint32)  # mx + 1, mx + 2: unknown ids
        sel[rng.random(c.P) < 0.01] = SEL_NEVER
        labs.append(col[None, :])
        sels.append(sel[None, :])
    lab, sel = np.concatenate(labs), np.concatenate(sels)
    if at is not None:  # move the last new column into position `at`
        order = list(range(lab.shape[0] - 1))
        order.insert(at, lab.shape[0] - 1)
        lab, sel = lab[order], sel[order]
    return np.ascontiguousarray(lab), np.ascontiguousarray(sel)


def run_case(ev, c, lab, sel, flags, taints=False, expect="fused"):
    tnt = c.node_taints if taints else None
    tol = c.pod_tol if taints else None
    ev.set_nodes(c.avail_cpu, c.avail_mem, lab, tnt)
    want = capi.eval_encoded(c.avail_cpu, c.avail_mem, lab, tnt, c.req_cpu, c.req_mem, sel, tol, c.samples, flags)
    for kernel in ("auto", "direct"):
        ev.set_kernel(kernel)
        r = ev.eval(c.req_cpu, c.req_mem, sel, tol, c.samples if flags & PICK_SAMPLED else None, flags)
        if kernel == "auto":
            assert ev.last_kernel == expect, (ev.last_kernel, expect)
        assert np.array_equal(r.feasible, want[0]), kernel
        if flags & WANT_FIT_MASK:
            assert np.array_equal(r.fit, want[1]), kernel
        if flags & (PICK_SAMPLED | PICK_BESTFIT):
            assert np.array_equal(r.binding, want[2]), kernel
    ev.set_kernel("auto")
    return want[0]


def test_hostname_key_beyond_the_first_eight_columns(evaluator):
    c = synth.make_cluster(6000, 5000, n_keys=8, n_taints=0, seed=21)
    lab, sel = with_keys(c, [5000], seed=1)
    feas = run_case(evaluator, c, lab, sel, FIT | SEL | PICK_SAMPLED)
    pinned = sel[8] != 0
    rows = np.unpackbits(feas.view(np.uint8), axis=1).sum(axis=1)
    assert pinned.sum() > 500 and rows[pinned].max() <= 1 and rows[~pinned].max() > 100  # a hostname selector leaves at most one node


def test_hostname_key_inside_the_first_eight_columns(evaluator):
    c = synth.make_cluster(3000, 5000, n_keys=8, n_taints=0, seed=22)
    lab, sel = with_keys(c, [5000], seed=2, at=3)
    run_case(evaluator, c, lab, sel, FIT | SEL | PICK_SAMPLED | WANT_FIT_MASK)


def test_two_list_keys_taints_fit_mask_bestfit_ragged(evaluator):
    c = synth.make_cluster(2500, 4321, n_keys=6, n_taints=16, seed=23)
    lab, sel = with_keys(c, [4321, 900], seed=3)
    run_case(evaluator, c, lab, sel, FIT | SEL | TAINT | WANT_FIT_MASK | PICK_BESTFIT, taints=True)


def test_mid_cardinality_long_ranges_take_the_checked_path(evaluator):
    """cardinality 40 on 2048 nodes: ~25 nodes per value and tile -> ranges longer than the 8 entries the unchecked path walks."""
    c = synth.make_cluster(2000, 2048, n_keys=8, n_taints=0, seed=24)
    # the eight synthetic keys fill 257 rows; 300 more rows do not fit next to them: the new key (and one more) become lists
    lab, sel = with_keys(c, [40, 300], seed=4)
    lab[8] = np.where(lab[8] == 0, 0, lab[8] * 9)  # spread the ids: the key's row count (max id) is what the layout sees
    sel[8] = np.where((sel[8] != 0) & (sel[8] != SEL_NEVER), sel[8] * 9, sel[8])
    run_case(evaluator, c, lab, sel, FIT | SEL | PICK_SAMPLED)


def test_more_than_eight_constrained_keys_with_a_list_key(evaluator):
    c = synth.make_cluster(1500, 3000, n_keys=8, n_taints=0, seed=25)
    lab, sel = with_keys(c, [3, 4, 5, 3000], p_constrain=0.6, seed=5)
    sel[:8][(sel[:8] == 0) & (np.random.default_rng(6).random((8, c.P)) < 0.7)] = 1  # most pods constrain most keys
    run_case(evaluator, c, lab, sel, FIT | SEL | PICK_SAMPLED)


@pytest.mark.parametrize("flags", [SEL, SEL | TAINT, FIT | SEL | WANT_FIT_MASK])
def test_predicate_subsets_with_a_list_key(evaluator, flags):
    c = synth.make_cluster(1200, 2600, n_keys=8, n_taints=8, seed=26)
    lab, sel = with_keys(c, [2600], seed=7)
    run_case(evaluator, c, lab, sel, flags, taints=bool(flags & TAINT))


def test_device_built_lists_equal_the_host_spec_and_survive_updates(evaluator):
    ev = evaluator
    c = synth.make_cluster(64, 5200, n_keys=8, n_taints=0, seed=27)
    lab, _ = with_keys(c, [5200, 700], seed=8)
    ev.set_nodes(c.avail_cpu, c.avail_mem, lab, None)
    dev = ev.index_checksum()
    ev.set_option(_lib.OPT_INDEX_BUILD, 1)
    try:
        ev.set_nodes(c.avail_cpu, c.avail_mem, lab, None)
        assert ev.index_checksum() == dev
    finally:
        ev.set_option(_lib.OPT_INDEX_BUILD, 0)
    ev.set_nodes(c.avail_cpu, c.avail_mem, lab, None)
    idx = np.array([5, 1024, 5199], dtype=np.uint32)
    ev.update_nodes(idx, c.avail_cpu[idx] - 1, c.avail_mem[idx] - 1)
    cpu, mem = c.avail_cpu.copy(), c.avail_mem.copy()
    cpu[idx] -= 1
    mem[idx] -= 1
    got = ev.index_checksum()
    ev.set_option(_lib.OPT_INDEX_BUILD, 1)
    try:
        ev.set_nodes(cpu, mem, lab, None)
        assert ev.index_checksum() == got
    finally:
        ev.set_option(_lib.OPT_INDEX_BUILD, 0)


def test_three_high_cardinality_keys_fall_back_with_a_reason(evaluator):
    c = synth.make_cluster(300, 2100, n_keys=8, n_taints=0, seed=28)
    lab, sel = with_keys(c, [2100, 2000, 1900], seed=9)
    run_case(evaluator, c, lab, sel, FIT | SEL, expect="direct")
    ev = evaluator
    ev.set_kernel("fused")
    with pytest.raises(Exception) as e:
        ev.eval(c.req_cpu, c.req_mem, sel, None, None, FIT | SEL)
    assert "high-cardinality" in str(e.value)
    ev.set_kernel("auto")


@pytest.mark.parametrize("n_taints", [0, 16])
def test_bestfit_on_a_snapshot_with_list_keys(evaluator, n_taints):
    """Best fit when some pods name a node (hostname key) or a mid-cardinality list key: those pods are picked from the key's sorted
    lists (k_pick_bestfit_listed: only the nodes carrying the value are candidates), the others from the bitmaps in best-fit order;
    no mask is read -- also as a bindings-only request, and again after the snapshot changed."""
    ev = evaluator
    c = synth.make_cluster(4000, 5500, n_keys=8, n_taints=n_taints, seed=29 + n_taints)
    lab, sel = with_keys(c, [5500, 700], p_constrain=0.3, seed=10)
    tnt = c.node_taints if n_taints else None
    tol = c.pod_tol if n_taints else None
    flags = FIT | SEL | (TAINT if n_taints else 0) | PICK_BESTFIT
    cpu, mem = c.avail_cpu.copy(), c.avail_mem.copy()
    ev.set_kernel("auto")
    ev.set_nodes(cpu, mem, lab, tnt)
    rng = np.random.default_rng(3)
    for step in range(2):
        _, _, want = capi.eval_encoded(cpu, mem, lab, tnt, c.req_cpu, c.req_mem, sel, tol, None, flags)
        got = ev.eval(c.req_cpu, c.req_mem, sel, tol, None, flags, want_mask=False)  # bindings only: no mask kernel at all
        assert np.array_equal(got.binding, want), step
        got = ev.eval(c.req_cpu, c.req_mem, sel, tol, None, flags)
        assert np.array_equal(got.binding, want), step
        pinned = (sel[8] != 0) & (sel[8] != SEL_NEVER)
        assert pinned.sum() > 500 and (want[pinned] >= 0).sum() > 50, "the case must exercise hostname-pinned pods that do get a node"
        idx = rng.choice(c.N, 40, replace=False).astype(np.uint32)
        cpu[idx] += rng.integers(-3000, 3000, idx.size)
        mem[idx] += rng.integers(-(1 << 30), 1 << 30, idx.size)
        ev.update_nodes(idx, cpu[idx], mem[idx])
