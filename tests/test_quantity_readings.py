"""Bracketing the UNPINNED half of the parity claim.

Resource fit runs on kube_quantity 0.6.1 in the reference (Cargo.lock:787-797), a crate that is not on disk and cannot be built
here.  oracle/oracle_ref.py therefore carries TWO readings of a quantity:
  exact      Kubernetes semantics, exact rationals (parse_quantity) -- what the product, oracle.c and every fixture implement;
  recalled   kube_quantity 0.6.1 as the surveyor recalls its internals (KubeQuantity061: Decimal values, scale / format
             conversions through f32 factors kept to 7 significant digits) -- a recollection, not a restatement of code on disk.
Where the two agree, a run of the real reference cannot tell them apart: the parity claim holds under either reading.  Where they
differ, the committed fixtures hold BOTH expectations (tests/golden/*_readings.json), so the day someone runs
`rust/pin_parity.sh <reference checkout>` the answer is a one-line diff (test_reference_run_says_which_reading below).

`pytest -s tests/test_quantity_readings.py` prints the per-spelling table.
"""
import json
import os

import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import pack_mask
from oracle import oracle_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["hazard_gi_24x10", "typical_specs_40x12"]


def frac(text):
    return None if text == "rejected" else R.Fraction(text)


def test_spelling_table_is_what_the_two_readings_say():
    doc = json.load(open(os.path.join(GOLD, "quantity_readings.json")))
    print("\n%-14s %-24s %-24s %s" % ("spelling", "exact Kubernetes", "kube_quantity 0.6.1 (recalled)", "agree"))
    for row in doc["rows"]:
        s = row["spelling"]
        try:
            exact = R.parse_quantity(s)
        except R.ReferencePanic:
            exact = None
        try:
            recalled = R.KubeQuantity061.parse(s).in_units()
        except R.ReferencePanic:
            recalled = None
        assert exact == frac(row["exact_kubernetes"]) and recalled == frac(row["kube_quantity_0_6_1_as_recalled"]), s
        assert row["agree"] == (exact == recalled)
        print("%-14s %-24s %-24s %s" % (s, row["exact_kubernetes"], row["kube_quantity_0_6_1_as_recalled"], "yes" if row["agree"] else "NO"))


def test_where_the_readings_agree_and_where_they_do_not():
    """Inside the parity claim (both readings agree): plain integers, decimal fractions, every DecimalSI suffix (n u m k M G T P E),
    Ki and Mi.  Outside it: Gi / Ti / Pi / Ei (the f32 factor does not survive 7 digits) and the exponent forms 0.6.1 does not parse."""
    rows = {r["spelling"]: r for r in json.load(open(os.path.join(GOLD, "quantity_readings.json")))["rows"]}
    inside = ["0", "1", "2", "500m", "250m", "100m", "10m", "1500m", "7910m", "0.5", "1.5", "64Mi", "128Mi", "512Mi", "768Mi", "1025Mi", "15031Mi", "196608Mi",
              "32779148Ki", "1G", "8G", "100M", "1k", "1T", "10E", "1073741824", "17179869184", "1n", "100u", "+5", "-5m"]
    outside = ["1Gi", "2Gi", "4Gi", "16Gi", "31Gi", "64Gi", "256Gi", "1.5Gi", "1Ti", "0.5Ti", "1Pi", "1e3", "1E3", "129e6", "5."]
    assert set(inside) | set(outside) == set(rows)
    assert all(rows[s]["agree"] for s in inside), [s for s in inside if not rows[s]["agree"]]
    assert not any(rows[s]["agree"] for s in outside), [s for s in outside if rows[s]["agree"]]
    # the size of the disagreement: Gi is inflated by 1.6e-7 (176 bytes per GiB), never more than 4e-7 for Ti / Pi
    for s in ("1Gi", "64Gi", "1.5Gi", "1Ti", "1Pi"):
        a, b = frac(rows[s]["exact_kubernetes"]), frac(rows[s]["kube_quantity_0_6_1_as_recalled"])
        assert 0 < (b - a) / a < R.Fraction(4, 10 ** 7), s


@pytest.mark.parametrize("name", CASES)
def test_both_expectations_are_committed_and_reproducible(name):
    objs = json.load(open(os.path.join(GOLD, name + "_objects.json")))
    doc = json.load(open(os.path.join(GOLD, name + "_readings.json")))
    P, N = doc["p"], doc["n"]
    _, fit = R.eval_matrix(objs["pods"], objs["nodes"], objs["bound"])
    kq = R.kq061_fit_matrix(objs["pods"], objs["nodes"], objs["bound"])
    hexrows = lambda bits: [[f"{int(w):016x}" for w in row] for row in pack_mask(np.array(bits, dtype=bool).reshape(P, N))]  # noqa: E731
    assert hexrows(fit) == doc["fit_exact_kubernetes"]
    assert hexrows([bool(x) for x in kq]) == doc["fit_kube_quantity_0_6_1_as_recalled"]
    assert [[i // N, i % N] for i, x in enumerate(kq) if x is None] == doc["pairs_the_recalled_parser_rejects"]
    # the exact reading is also what the *_expected.json fixture (and the product, tests/test_gpu_objects.py) holds
    assert doc["fit_exact_kubernetes"] == json.load(open(os.path.join(GOLD, name + "_expected.json")))["fit"]
    print(f"\n{name}: the readings differ on {len(doc['pairs_where_the_readings_differ'])} of {P * N} (pod, node) pairs")
    for p, n in doc["pairs_where_the_readings_differ"]:
        req = objs["pods"][p]["spec"]["containers"][0]["resources"]["requests"]
        al = objs["nodes"][n]["status"]["allocatable"]
        exact = (int(doc["fit_exact_kubernetes"][p][n >> 6], 16) >> (n & 63)) & 1
        print(f"  pod {p:2d} requests {req}  node {n:2d} allocatable {al}: exact {'fits' if exact else 'does not fit'}, recalled reading says the opposite")


def test_typical_specs_disagree_only_at_exact_fit_boundaries_of_gi_quantities():
    """On typical real spellings the two readings give the same fit bit everywhere except where a Gi-or-larger quantity meets a byte
    count spelled another way within 4e-7 of it (a pod that fits its node EXACTLY): 3 of 480 pairs in this fixture."""
    objs = json.load(open(os.path.join(GOLD, "typical_specs_40x12_objects.json")))
    doc = json.load(open(os.path.join(GOLD, "typical_specs_40x12_readings.json")))
    assert doc["pairs_the_recalled_parser_rejects"] == []
    assert 0 < len(doc["pairs_where_the_readings_differ"]) <= 6
    for p, n in doc["pairs_where_the_readings_differ"]:
        req = objs["pods"][p]["spec"]["containers"][0]["resources"]["requests"]["memory"]
        al = objs["nodes"][n]["status"]["allocatable"]["memory"]
        assert any(s.endswith(("Gi", "Ti")) for s in (req, al)), (req, al)
        a, b = R.parse_quantity(req), R.parse_quantity(al)
        bound = sum((R.total_pod_resources(q).memory for q in R.list_pods_on_node(objs["bound"], objs["nodes"][n]["metadata"]["name"])), R.Fraction(0))
        assert abs(a - (b - bound)) <= R.Fraction(4, 10 ** 7) * b, "only an (almost) exact fit can flip"


@pytest.mark.parametrize("name", CASES)
def test_reference_run_says_which_reading(name):
    """With tests/golden/ref_<name>.json (written by rust/pin_parity.sh on a box with cargo): which reading is the reference's?"""
    path = os.path.join(GOLD, f"ref_{name}.json")
    if not os.path.exists(path):
        pytest.skip(f"tests/golden/ref_{name}.json absent: no cargo here or on the GPU box; `rust/pin_parity.sh <reference checkout>` writes it")
    ref = json.load(open(path))["fit"]
    doc = json.load(open(os.path.join(GOLD, name + "_readings.json")))
    is_exact, is_recalled = ref == doc["fit_exact_kubernetes"], ref == doc["fit_kube_quantity_0_6_1_as_recalled"]
    print(f"\n{name}: the reference's fit mask equals the exact reading: {is_exact}; equals the recalled kube_quantity reading: {is_recalled}")
    assert is_exact or is_recalled, "the reference follows NEITHER reading: the recollection of kube_quantity 0.6.1 is wrong somewhere -- read the crate"
