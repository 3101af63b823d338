"""Known-answer tests of the oracle.

KAT-S1..S3 are the reference's own tests (src/predicates/test.rs:42-58), replayed with the same
fixtures (test.rs:13-40).  D-* are the derived vectors of SURVEY.md section 8c, each following
from a cited line of the reference.  Both restatements (oracle_ref.py and oracle.c) must agree.
"""
import pytest

from oracle import capi, oracle_ref as R

POD_NAMESPACE, POD_NAME, NODE_NAME = "test", "pod1", "node1"  # src/predicates/test.rs:8-11


def make_test_pod(selector_key=None):
    """test_pod fixture, src/predicates/test.rs:13-28: spec is None unless a selector is given."""
    pod = {"metadata": {"namespace": POD_NAMESPACE, "name": POD_NAME}}
    if selector_key is not None:
        k, v = selector_key
        pod["spec"] = {"nodeSelector": {k: v}}
    return pod


def make_test_node():
    """test_node fixture, src/predicates/test.rs:30-40: labels {name: node1}, no status."""
    return {"metadata": {"name": NODE_NAME, "labels": {"name": NODE_NAME}}}


IMPLS = [pytest.param(R.does_node_selector_match, id="oracle_ref.py"),
         pytest.param(capi.does_node_selector_match, id="oracle.c")]


@pytest.mark.parametrize("match", IMPLS)
def test_does_node_selector_match_no_selector(match):  # KAT-S1, test.rs:42-45
    assert match(make_test_pod(), make_test_node()) is True


@pytest.mark.parametrize("match", IMPLS)
def test_does_node_selector_match_false(match):  # KAT-S2, test.rs:47-50
    assert match(make_test_pod(("foo", "bar")), make_test_node()) is False


@pytest.mark.parametrize("match", IMPLS)
def test_does_node_selector_match_true(match):  # KAT-S3, test.rs:52-58
    assert match(make_test_pod(("name", NODE_NAME)), make_test_node()) is True


# ---- derived selector vectors ----------------------------------------------------------------------
def _pod_sel(sel):
    return {"metadata": {"name": "p"}, "spec": {"nodeSelector": sel}}


def _node_lab(labels):
    md = {"name": "n"}
    if labels is not None:
        md["labels"] = labels
    return {"metadata": md}


SEL_VECTORS = [
    ("D-S4 empty selector map", _pod_sel({}), _node_lab({"a": "b"}), True),
    ("D-S4 empty selector, node labels None", _pod_sel({}), _node_lab(None), True),
    ("D-S5 selector, node labels None", _pod_sel({"a": "b"}), _node_lab(None), False),
    ("D-S6 no selector, node labels None", {"metadata": {"name": "p"}, "spec": {}}, _node_lab(None), True),
    ("D-S7 wrong value", _pod_sel({"name": "other"}), _node_lab({"name": "node1"}), False),
    ("D-S8 two keys, superset labels", _pod_sel({"a": "1", "b": "2"}), _node_lab({"a": "1", "b": "2", "c": "3"}), True),
    ("D-S8 two keys, one differs", _pod_sel({"a": "1", "b": "2"}), _node_lab({"a": "1", "b": "9"}), False),
    ("D-S9 empty string value present", _pod_sel({"a": ""}), _node_lab({"a": ""}), True),
    ("D-S9 empty string value, key absent", _pod_sel({"a": ""}), _node_lab({"b": ""}), False),
    ("case sensitive", _pod_sel({"a": "B"}), _node_lab({"a": "b"}), False),
    ("prefix is not equality", _pod_sel({"a": "bb"}), _node_lab({"a": "b"}), False),
]


@pytest.mark.parametrize("match", IMPLS)
@pytest.mark.parametrize("name,pod,node,want", SEL_VECTORS, ids=[v[0] for v in SEL_VECTORS])
def test_selector_derived(match, name, pod, node, want):
    assert match(pod, node) is want


# ---- derived resource vectors ----------------------------------------------------------------------
MIB, GIB = 1 << 20, 1 << 30


def _cont(cpu=None, mem=None):
    req = {}
    if cpu is not None:
        req["cpu"] = cpu
    if mem is not None:
        req["memory"] = mem
    return {"name": "c", "resources": {"requests": req}}


def _pod(conts, node_name=None, extra=None):
    spec = {"containers": conts}
    if node_name:
        spec["nodeName"] = node_name
    if extra:
        spec.update(extra)
    return {"metadata": {"name": "p", "namespace": "ns"}, "spec": spec}


def _node(cpu="1", mem=str(GIB), name="node1", status=True):
    n = {"metadata": {"name": name, "labels": {}}}
    if status:
        n["status"] = {"allocatable": {"cpu": cpu, "memory": mem}}
    return n


def _both_check(pod, node, all_pods):
    """reason from both restatements; they must agree."""
    r_py = R.check_node_validity(pod, node, R.list_pods_on_node(all_pods, R.node_name(node)))
    r_c = capi.check_node_validity(pod, node, all_pods)
    names = {0: None, 1: "NotEnoughResources", 2: "NodeSelectorMismatch"}
    assert names[r_c] == r_py
    return r_py


def test_D_R1_simple_fit():
    assert _both_check(_pod([_cont("500m", str(128 * MIB))]), _node("1", str(GIB)), []) is None


def test_D_R2_exact_fit_is_feasible():  # <=, src/predicates.rs:42
    assert _both_check(_pod([_cont("1000m", str(GIB))]), _node("1", str(GIB)), []) is None


def test_D_R3_one_byte_over():
    assert _both_check(_pod([_cont("1", str(GIB + 1))]), _node("1", str(GIB)), []) == "NotEnoughResources"


def test_D_R4_overcommitted_node_rejects_zero_request_pod():  # available negative, src/predicates.rs:37
    bound = [_pod([_cont("2", str(GIB))], node_name="node1")]
    assert _both_check(_pod([{"name": "c"}]), _node("1", str(GIB)), bound) == "NotEnoughResources"


def test_D_R5_no_status_zero_request():  # available stays 0/0, src/predicates.rs:27-32
    assert _both_check(_pod([{"name": "c"}]), _node(status=False), []) is None
    assert _both_check(_pod([_cont("1m", "0")]), _node(status=False), []) == "NotEnoughResources"


def test_D_R6_container_sum():  # src/util.rs:58-69
    pod = _pod([_cont("250m", "100"), _cont("250m", None), {"name": "no-resources"}, {"name": "r", "resources": {}}])
    res = R.total_pod_resources(pod)
    assert res.cpu * 1000 == 500 and res.memory == 100
    assert capi.total_pod_resources(pod) == (500 * 10**6, 100 * 10**9)
    assert _both_check(pod, _node("500m", "100"), []) is None
    assert _both_check(pod, _node("499m", "100"), []) == "NotEnoughResources"


def test_D_R7_init_containers_and_limits_ignored():  # src/util.rs:58
    pod = _pod([_cont("100m", "10")], extra={"initContainers": [_cont("64", "1Ti")], "overhead": {"cpu": "64"}})
    pod["spec"]["containers"][0]["resources"]["limits"] = {"cpu": "64", "memory": "1Ti"}
    assert _both_check(pod, _node("100m", "10"), []) is None


def test_D_R8_succeeded_pod_still_counts():  # no phase filter, src/predicates.rs:22-25,36-38
    done = _pod([_cont("600m", "1")], node_name="node1")
    done["status"] = {"phase": "Succeeded"}
    other = _pod([_cont("600m", "1")], node_name="node2")
    assert _both_check(_pod([_cont("500m", "1")]), _node("1", "10"), [done, other]) == "NotEnoughResources"
    assert _both_check(_pod([_cont("400m", "1")]), _node("1", "10"), [done, other]) is None


def test_D_V1_both_fail_reports_resources_first():  # src/predicates.rs:68-70
    pod = _pod([_cont("2", "1")], extra={"nodeSelector": {"a": "b"}})
    assert _both_check(pod, _node("1", "10"), []) == "NotEnoughResources"


def test_D_V2_selector_only_fails():  # src/predicates.rs:72-74
    pod = _pod([_cont("1", "1")], extra={"nodeSelector": {"a": "b"}})
    assert _both_check(pod, _node("1", "10"), []) == "NodeSelectorMismatch"


def test_allocatable_missing_key_panics():  # src/predicates.rs:29-31 BTreeMap index
    node = {"metadata": {"name": "n"}, "status": {"allocatable": {"cpu": "1"}}}
    with pytest.raises(R.ReferencePanic):
        R.can_pod_fit(_pod([]), node, [])
    assert capi.check_node_validity(_pod([]), node, []) == capi.E_MISSING_KEY


def test_bad_quantity_panics():  # src/util.rs:65 expect
    with pytest.raises(R.ReferencePanic):
        R.total_pod_resources(_pod([_cont("abc", "1")]))
    assert capi.check_node_validity(_pod([_cont("abc", "1")]), _node(), []) == capi.E_PARSE


# ---- pick -------------------------------------------------------------------------------------------
def _nodes8():
    # only node 7 is big enough
    return [_node("1" if i != 7 else "8", str(GIB), name=f"node{i}") for i in range(8)]


def test_D_P1_third_attempt_wins():  # src/main.rs:53-66
    pod = _pod([_cont("4", "1")])
    samples = [3, 3, 7, 1, 0]
    assert R.select_node_for_pod(pod, _nodes8(), [], samples) == 7
    assert capi.select_node_for_pod(pod, _nodes8(), [], samples) == 7


def test_D_P2_zero_nodes():  # src/main.rs:56,70
    pod = _pod([_cont("4", "1")])
    assert R.select_node_for_pod(pod, [], [], [0, 0, 0, 0, 0]) is None
    assert capi.select_node_for_pod(pod, [], [], [0, 0, 0, 0, 0]) == -1


def test_D_P3_feasible_node_exists_but_not_drawn():  # src/main.rs:53-70,117
    pod = _pod([_cont("4", "1")])
    samples = [0, 1, 2, 3, 4]
    assert R.select_node_for_pod(pod, _nodes8(), [], samples) is None
    assert capi.select_node_for_pod(pod, _nodes8(), [], samples) == -1


def test_pick_first_feasible_not_best():
    pod = _pod([_cont("1", "1")])
    nodes = [_node("8", str(GIB), name=f"node{i}") for i in range(4)]
    assert R.select_node_for_pod(pod, nodes, [], [2, 0, 1, 3, 3]) == 2
    assert capi.select_node_for_pod(pod, nodes, [], [2, 0, 1, 3, 3]) == 2


# ---- quantities ---------------------------------------------------------------------------------------
QUANTITIES = [("0", 0), ("1", 10**9), ("250m", 25 * 10**7), ("1500m", 15 * 10**8), ("1.5", 15 * 10**8), ("100n", 100),
              ("1k", 10**12), ("1M", 10**15), ("1Ki", 1024 * 10**9), ("1Mi", MIB * 10**9), ("1Gi", GIB * 10**9),
              ("1Ti", (1 << 40) * 10**9), ("129e6", 129 * 10**15), ("1E3", 10**12), ("1E", 10**27), ("+5", 5 * 10**9),
              ("-5", -5 * 10**9), ("123456789012", 123456789012 * 10**9), (".5", 5 * 10**8), ("5.", 5 * 10**9),
              ("0.001", 10**6), ("12e-3", 12 * 10**6)]


@pytest.mark.parametrize("text,nanos", QUANTITIES, ids=[q[0] for q in QUANTITIES])
def test_quantity_values(text, nanos):
    assert R.parse_quantity(text) * 10**9 == nanos
    assert capi.parse_quantity(text) == nanos


@pytest.mark.parametrize("text", ["", "abc", "1Zi", "1mm", "m", ".", "1 ", " 1", "1e", "--1", "1.2.3", "Ki"])
def test_quantity_rejects(text):
    with pytest.raises(R.ReferencePanic):
        R.parse_quantity(text)
    with pytest.raises(ValueError):
        capi.parse_quantity(text)


# ---- taints / tolerations (extension E2) ------------------------------------------------------------------
def _tnode(taints):
    return {"metadata": {"name": "n"}, "spec": {"taints": taints}}


def _tpod(tols):
    return {"metadata": {"name": "p"}, "spec": {"tolerations": tols}}


TAINT_VECTORS = [
    ("no taints", _tpod([]), _tnode([]), True),
    ("untolerated", _tpod([]), _tnode([{"key": "a", "value": "1", "effect": "NoSchedule"}]), False),
    ("equal match", _tpod([{"key": "a", "operator": "Equal", "value": "1", "effect": "NoSchedule"}]),
     _tnode([{"key": "a", "value": "1", "effect": "NoSchedule"}]), True),
    ("default operator is Equal", _tpod([{"key": "a", "value": "1"}]), _tnode([{"key": "a", "value": "1", "effect": "NoSchedule"}]), True),
    ("value differs", _tpod([{"key": "a", "value": "2"}]), _tnode([{"key": "a", "value": "1", "effect": "NoSchedule"}]), False),
    ("exists ignores value", _tpod([{"key": "a", "operator": "Exists"}]), _tnode([{"key": "a", "value": "1", "effect": "NoExecute"}]), True),
    ("effect differs", _tpod([{"key": "a", "operator": "Exists", "effect": "NoExecute"}]),
     _tnode([{"key": "a", "value": "1", "effect": "NoSchedule"}]), False),
    ("empty key exists tolerates all", _tpod([{"operator": "Exists"}]),
     _tnode([{"key": "a", "value": "1", "effect": "NoSchedule"}, {"key": "b", "effect": "NoExecute"}]), True),
    ("prefer-no-schedule never filters", _tpod([]), _tnode([{"key": "a", "effect": "PreferNoSchedule"}]), True),
    ("one of two untolerated", _tpod([{"key": "a", "operator": "Exists"}]),
     _tnode([{"key": "a", "effect": "NoSchedule"}, {"key": "b", "effect": "NoSchedule"}]), False),
]


@pytest.mark.parametrize("name,pod,node,want", TAINT_VECTORS, ids=[v[0] for v in TAINT_VECTORS])
def test_taints(name, pod, node, want):
    assert R.tolerates_node_taints(pod, node) is want
    assert capi.tolerates_node_taints(pod, node) is want
