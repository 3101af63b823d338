"""Pinning to the REFERENCE's own predicates (the step VERDICT r1 calls "make fit pinnable in one command").

tests/golden/<name>_objects.json are Kubernetes JSON objects of seeded clusters (made by tests/golden/make_golden.py).
A maintainer with cargo runs ONE command,

    rust/pin_parity.sh /path/to/kube-scheduler-rs-reference

which overlays rust/ on a copy of the reference, runs the reference's own `does_node_selector_match` and the pure half of
`can_pod_fit` (src/predicates.rs:20-61, arithmetic in kube_quantity 0.6.1) on those objects and writes
tests/golden/ref_<name>.json.  This module then compares them with the committed fixtures:

  * domain D cases (c1_100x20, ragged_70x130_taints, one_node_33x1, binsuffix_60x40): every fit / selector / feasible word and
    every sampled pick must be identical -> resource-fit parity is PINNED;
  * hazard_gi_24x10 (Gi / Ti / exponent spellings, outside D): differences are reported as the documented divergence
    (kube_quantity's suspected f32 scale conversion vs exact Kubernetes semantics), xfail, not a parity failure.

Neither this container nor the GPU box has cargo (profiles/r02_a_toolchain_probe_gpu_box.txt), so without ref_*.json these
tests SKIP with that reason -- the oracle's header keeps saying "parity unpinned" for fit until they run.

What always runs here (no reference needed): the exported objects describe the same clusters as the encoded fixtures, and the
C object-level oracle reproduces the object-only expectations (binary suffixes, hazard spellings).
"""
import glob
import json
import os

import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import pack_mask
from oracle import capi
from oracle import oracle_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
OBJECTS = sorted(glob.glob(os.path.join(GOLD, "*_objects.json")))
NAMES = [os.path.basename(p)[: -len("_objects.json")] for p in OBJECTS]
HAZARD = {"hazard_gi_24x10", "typical_specs_40x12", "subunit_22x8"}  # quantities outside domain D (Gi and above): tests/test_quantity_readings.py holds both expectations
SKIP_REASON = ("tests/golden/ref_{name}.json absent: produced by the reference itself via `rust/pin_parity.sh <reference checkout>` "
               "on a box with cargo (none here, none on the GPU box); resource-fit parity stays UNPINNED until it exists")


def unhex(rows):
    return np.array([[int(w, 16) for w in r] for r in rows], dtype=np.uint64).reshape(len(rows), -1)


def load(name):
    return json.load(open(os.path.join(GOLD, name + "_objects.json")))


def expected(name):
    """(fit, sel, fit&sel, sampled) of the committed fixtures for `name`."""
    npz = os.path.join(GOLD, name + ".npz")
    doc = load(name)
    P, N = doc["p"], doc["n"]
    if os.path.exists(npz):
        g = np.load(npz)
        sel, _, _ = capi.eval_encoded(g["avail_cpu"], g["avail_mem"], g["node_labels"], None, g["req_cpu"], g["req_mem"], g["pod_sel"], None, None, capi.SEL)
        _, _, smp = capi.eval_encoded(g["avail_cpu"], g["avail_mem"], g["node_labels"], None, g["req_cpu"], g["req_mem"], g["pod_sel"], None,
                                      g["samples"], capi.FIT | capi.SEL | capi.PICK_SAMPLED)
        return g["fit"], sel.reshape(P, -1), g["fit"] & sel.reshape(P, -1), smp
    e = json.load(open(os.path.join(GOLD, name + "_expected.json")))
    return unhex(e["fit"]), unhex(e["sel"]), unhex(e["feasible_fit_and_sel"]), np.array(e["sampled"], dtype=np.int32)


def test_object_fixtures_present():
    assert {"c1_100x20", "ragged_70x130_taints", "one_node_33x1", "binsuffix_60x40", "hazard_gi_24x10"} <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_objects_describe_the_fixture_cluster(name):
    """The exported objects, evaluated one pair at a time by the C object-level oracle (strings and maps), give the
    fixture's masks: the JSON a maintainer feeds to the reference IS the cluster the encoded fixtures describe."""
    doc = load(name)
    fit, sel, both, smp = expected(name)
    feas_o, fit_o = capi.eval_objects(doc["pods"], doc["nodes"], doc["bound"], capi.FIT | capi.SEL, want_fit=True)
    assert np.array_equal(fit_o, fit)
    assert np.array_equal(feas_o, both)
    sel_o, _ = capi.eval_objects(doc["pods"], doc["nodes"], doc["bound"], capi.SEL)
    assert np.array_equal(sel_o, sel)
    picks = [capi.select_node_for_pod(p, doc["nodes"], doc["bound"], doc["samples"][i]) for i, p in enumerate(doc["pods"][:40])]
    assert picks == [int(x) for x in smp[:40]]
    assert [n["metadata"]["name"] for n in doc["nodes"]] == sorted(n["metadata"]["name"] for n in doc["nodes"]), "canonical order"


def test_hazard_expectations_follow_kubernetes_semantics():
    """The hazard case's expectations are exact powers of 1024 / 1000: spot values the f32 reading would get wrong."""
    assert R.parse_quantity("1Gi") == 1 << 30 and R.parse_quantity("1.5Gi") == 3 << 29 and R.parse_quantity("1e10") == 10 ** 10
    doc = load("hazard_gi_24x10")
    fit, _, _, _ = expected("hazard_gi_24x10")
    bit = lambda p, n: (int(fit[p, n >> 6]) >> (n & 63)) & 1  # noqa: E731
    assert bit(0, 0) == 1   # 1073741824 bytes on a "1Gi" node: exact fit
    assert bit(1, 0) == 0   # one byte more: no fit
    assert bit(2, 0) == 1 and bit(3, 0) == 1 and bit(4, 0) == 1  # 1Gi == 1024Mi == 1048576Ki
    assert doc["domain"].startswith("OUTSIDE D")


@pytest.mark.parametrize("name", NAMES)
def test_reference_outputs_equal_fixtures(name):
    path = os.path.join(GOLD, f"ref_{name}.json")
    if not os.path.exists(path):
        pytest.skip(SKIP_REASON.format(name=name))
    ref = json.load(open(path))
    fit, sel, both, smp = expected(name)
    diffs = []
    for key, want in (("fit", fit), ("sel", sel), ("feasible_fit_and_sel", both)):
        got = unhex(ref[key])
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            diffs.append(f"{key}: {len(bad)} words differ, first at pod {bad[0][0]} word {bad[0][1]}: reference {int(got[tuple(bad[0])]):#x} != fixture {int(want[tuple(bad[0])]):#x}")
    if "sampled" in ref and [int(x) for x in ref["sampled"]] != [int(x) for x in smp]:
        diffs.append("sampled picks differ")
    if ref.get("panics"):
        diffs.append(f"reference panicked on {len(ref['panics'])} pairs: {ref['panics'][:3]}")
    if name in HAZARD:
        if diffs:
            pytest.xfail("documented divergence outside domain D (kube_quantity scale conversion): " + "; ".join(diffs))
        return
    assert not diffs, "REFERENCE DIVERGENCE on the parity domain: " + "; ".join(diffs)
