"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the exact
object-level restatement).  CPU: the C oracle reproduces them.  GPU: so does the HIP path."""
import glob
import os

import numpy as np
import pytest

from oracle import capi

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _flags(g):
    return capi.FIT | capi.SEL | (capi.TAINT if int(g["n_taints"]) > 0 else 0)


def test_fixtures_present():
    assert len(FIXTURES) >= 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_c_oracle_reproduces_golden(path):
    g = np.load(path)
    fl = _flags(g)
    taints = g["node_taints"] if int(g["n_taints"]) else None
    tol = g["pod_tol"] if int(g["n_taints"]) else None
    feas, fit, bb = capi.eval_encoded(g["avail_cpu"], g["avail_mem"], g["node_labels"], taints, g["req_cpu"], g["req_mem"],
                                      g["pod_sel"], tol, None, fl | capi.WANT_FIT_MASK | capi.PICK_BESTFIT)
    assert np.array_equal(feas, g["feasible"])
    assert np.array_equal(fit, g["fit"])
    assert np.array_equal(bb, g["bestfit"])
    if "sampled" in g:
        _, _, bs = capi.eval_encoded(g["avail_cpu"], g["avail_mem"], g["node_labels"], taints, g["req_cpu"], g["req_mem"],
                                     g["pod_sel"], tol, g["samples"], fl | capi.PICK_SAMPLED)
        assert np.array_equal(bs, g["sampled"])


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["direct", "fused"])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_hip_reproduces_golden(evaluator, path, kernel):
    from kube_scheduler_rs_reference_amd import FIT, SEL, TAINT, PICK_BESTFIT, PICK_SAMPLED, WANT_FIT_MASK
    g = np.load(path)
    nt = int(g["n_taints"])
    ev = evaluator
    ev.set_kernel(kernel)
    ev.set_nodes(g["avail_cpu"], g["avail_mem"], g["node_labels"], g["node_taints"] if nt else None)
    fl = FIT | SEL | (TAINT if nt else 0)
    tol = g["pod_tol"] if nt else None
    r = ev.eval(g["req_cpu"], g["req_mem"], g["pod_sel"], tol, None, fl | WANT_FIT_MASK | PICK_BESTFIT)
    assert np.array_equal(r.feasible, g["feasible"])
    assert np.array_equal(r.fit, g["fit"])
    assert np.array_equal(r.binding, g["bestfit"])
    if "sampled" in g:
        r = ev.eval(g["req_cpu"], g["req_mem"], g["pod_sel"], tol, g["samples"], fl | PICK_SAMPLED)
        assert np.array_equal(r.binding, g["sampled"])
    ev.set_kernel("auto")
