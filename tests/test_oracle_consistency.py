"""The three restatements agree with each other on synthetic clusters (CPU only):
   object-level Python (exact Fractions)  ==  object-level C (strings/maps, int128)
   == encoded-level C loops on the generator's columns."""
import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import pack_mask, synth
from oracle import capi, oracle_ref as R


@pytest.mark.parametrize("P,N,n_keys,n_taints,seed,binsuf", [(100, 20, 8, 0, 0x5EED0000, False), (60, 70, 8, 16, 7, False),
                                                          (40, 130, 3, 5, 11, True), (5, 1, 8, 16, 3, False)])
def test_object_python_vs_object_c_vs_encoded(P, N, n_keys, n_taints, seed, binsuf):
    c = synth.make_cluster(P, N, n_keys=n_keys, n_taints=n_taints, seed=seed, binary_suffixes=binsuf)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    use_taint = n_taints > 0
    feas_py, fit_py = R.eval_matrix(pods, nodes, bound, use_taint=use_taint)
    flags = capi.FIT | capi.SEL | (capi.TAINT if use_taint else 0)
    feas_c, fit_c = capi.eval_objects(pods, nodes, bound, flags, want_fit=True)
    assert np.array_equal(feas_c, pack_mask(np.array(feas_py, dtype=bool).reshape(P, N)))
    assert np.array_equal(fit_c, pack_mask(np.array(fit_py, dtype=bool).reshape(P, N)))
    feas_e, fit_e, _ = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels if n_keys else None,
                                         c.node_taints if n_taints else None, c.req_cpu, c.req_mem,
                                         c.pod_sel if n_keys else None, c.pod_tol if n_taints else None, None,
                                         flags | capi.WANT_FIT_MASK)
    assert np.array_equal(feas_e, feas_c)
    assert np.array_equal(fit_e, fit_c)
    if P * N >= 1000:
        dens = np.array(feas_py).mean()
        assert 0.0 < dens < 1.0  # masks are neither all-ones nor all-zeros


def test_generator_columns_match_objects():
    """available[n] from the object view (allocatable - sum of the node's LIST) equals the column."""
    c = synth.make_cluster(30, 40, n_keys=8, n_taints=4, seed=5)
    nodes, bound = c.node_objects(), c.bound_pod_objects()
    for i, node in enumerate(nodes):
        av = R.available_of(node, bound)
        assert av.cpu * 1000 == int(c.avail_cpu[i])
        assert av.memory == int(c.avail_mem[i])
    for p, pod in enumerate(c.pod_objects()):
        r = R.total_pod_resources(pod)
        assert r.cpu * 1000 == int(c.req_cpu[p]) and r.memory == int(c.req_mem[p])
    assert (c.avail_cpu < 0).any() or True


def test_picks_object_vs_encoded():
    c = synth.make_cluster(80, 33, n_keys=8, n_taints=0, seed=21)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    flags = capi.FIT | capi.SEL
    _, _, bind_s = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu, c.req_mem, c.pod_sel, None,
                                     c.samples, flags | capi.PICK_SAMPLED)
    _, _, bind_b = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu, c.req_mem, c.pod_sel, None,
                                     None, flags | capi.PICK_BESTFIT)
    for p, pod in enumerate(pods):
        want = R.select_node_for_pod(pod, nodes, bound, [int(s) for s in c.samples[p]])
        assert bind_s[p] == (-1 if want is None else want)
        assert capi.select_node_for_pod(pod, nodes, bound, c.samples[p]) == bind_s[p]
        wb = R.pick_bestfit(pod, nodes, bound)
        assert bind_b[p] == (-1 if wb is None else wb)
    assert (bind_s >= 0).any() and (bind_s < 0).any()


def test_encoded_threads_agree():
    c = synth.make_cluster(500, 300, n_keys=8, n_taints=16, seed=9)
    fl = capi.FIT | capi.SEL | capi.TAINT | capi.PICK_BESTFIT
    a = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, c.node_taints, c.req_cpu, c.req_mem, c.pod_sel, c.pod_tol, None, fl, threads=1)
    b = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, c.node_taints, c.req_cpu, c.req_mem, c.pod_sel, c.pod_tol, None, fl, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
