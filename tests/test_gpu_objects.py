"""The PRODUCT's object -> column path on whole clusters, against the independent object-level oracle.

tests/cpp/objects_eval (C++: tests/cpp/objects_eval.cpp) loads Kubernetes JSON objects, runs them through the host mirror --
host/quantity.cpp (quantity strings -> exact i64), host/encoder.cpp (canonical order, `available` from the LISTs, label
interning, taint bits), host/predicates.cpp / host/scheduler.cpp -- and the device (ksched_set_nodes, ksched_eval), and
prints what came back.  The expectation is oracle/oracle_ref.py: a different parser (regex + Fraction), dict lookups, one
per-pair evaluation at a time.  (oracle.c has its own, third parser -- a table-driven recogniser + rational assembly -- checked against
this one in tests/test_oracle_consistency.py.)

  * masks: the five golden object sets + a 2000 x 500 cluster with Ki/Mi spellings, 12 label keys (> 8: the kernel's overflow walk)
    and 16 taints  (row a3 of SURVEY.md section 8);
  * reconcile_batch == the oracle's restatement of the batched reference execution (row f-n2);
  * reconcile_batch_sequential == the oracle's restatement of the round-based accounting (row f-n3), including the snapshot the
    device is left with (incremental ksched_update_nodes == re-LIST).
"""
import json
import os
import subprocess

import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import pack_mask, synth
from oracle import oracle_ref as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(ROOT, "tests", "cpp", "objects_eval")


def tool(*args):
    if os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-C", ROOT, "-s", "host"])
    r = subprocess.run([TOOL, *map(str, args)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout)


def unhex(rows, W):
    return np.array([[int(w, 16) for w in r] for r in rows], dtype=np.uint64).reshape(len(rows), W)


def write_objects(path, pods, nodes, bound):
    with open(path, "w") as f:
        json.dump({"name": "tmp", "pods": pods, "nodes": nodes, "bound": bound, "samples": []}, f)


def expect_masks(pods, nodes, bound, use_taint, cache):
    P, N = len(pods), len(nodes)
    feas, fit = R.eval_matrix(pods, nodes, bound, use_taint=use_taint, cache=cache)
    return pack_mask(np.array(feas, dtype=bool).reshape(P, N)), pack_mask(np.array(fit, dtype=bool).reshape(P, N))


@pytest.mark.parametrize("name,taints", [("c1_100x20", False), ("ragged_70x130_taints", True), ("one_node_33x1", True),
                                         ("binsuffix_60x40", False), ("hazard_gi_24x10", False), ("subunit_22x8", False),
                                         ("wide_selectors_48x90", False)])
def test_golden_objects_through_the_host_encoder(name, taints):
    path = os.path.join(GOLD, name + "_objects.json")
    doc = json.load(open(path))
    got = tool("masks", path, *(["taints"] if taints else []))
    W = (doc["n"] + 63) // 64
    feas, fit = expect_masks(doc["pods"], doc["nodes"], doc["bound"], taints, cache=False)  # every pair re-parsed, as the reference does
    assert got["names"] == [n["metadata"]["name"] for n in doc["nodes"]], "canonical order = ascending node name"
    assert np.array_equal(unhex(got["fit"], W), fit)
    assert np.array_equal(unhex(got["feasible"], W), feas)
    assert got["list_calls"] == doc["n"], "one LIST per node per batch (not per evaluation, src/predicates.rs:34)"


def test_cluster_2000x500_binary_suffixes_12_keys_taints(tmp_path):
    c = synth.make_cluster(P=2000, N=500, n_keys=12, n_taints=16, seed=0x0B1EC7, binary_suffixes=True)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    assert max(len((p["spec"].get("nodeSelector") or {})) for p in pods) > 4
    path = tmp_path / "objs.json"
    write_objects(path, pods, nodes, bound)
    got = tool("masks", path, "taints")
    W = (c.N + 63) // 64
    feas, fit = expect_masks(pods, nodes, bound, True, cache=True)
    assert np.array_equal(unhex(got["fit"], W), fit)
    assert np.array_equal(unhex(got["feasible"], W), feas)
    dens = np.unpackbits(feas.view(np.uint8)).sum() / (c.P * c.N)
    assert 0.02 < dens < 0.9


def test_more_than_32_selector_keys_in_one_batch(tmp_path):
    """The device takes 32 label columns per call; the reference has no such limit (src/predicates.rs:45-61 walks any map).
    A batch whose pods use 45 distinct keys is evaluated in pod ranges, each within the budget -- same masks."""
    c = synth.make_cluster(P=90, N=70, n_keys=4, n_taints=0, seed=0x4B45)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    for i, n in enumerate(nodes):
        n["metadata"].setdefault("labels", {})
        for k in range(45):
            if (i + k) % 3:
                n["metadata"]["labels"][f"extra{k:02d}"] = f"v{(i * 7 + k) % 2}"
    for i, p in enumerate(pods):
        sel = p["spec"].setdefault("nodeSelector", {})
        sel[f"extra{i % 45:02d}"] = f"v{i % 2}"
        if i % 5 == 0:
            sel[f"extra{(i + 11) % 45:02d}"] = "v0"
    path = tmp_path / "objs.json"
    write_objects(path, pods, nodes, bound)
    got = tool("masks", path)
    W = (c.N + 63) // 64
    feas, fit = expect_masks(pods, nodes, bound, False, cache=True)
    assert np.array_equal(unhex(got["fit"], W), fit)
    assert np.array_equal(unhex(got["feasible"], W), feas)
    assert 0 < np.unpackbits(feas.view(np.uint8)).sum() < np.unpackbits(fit.view(np.uint8)).sum()


def test_a_pod_with_forty_selector_keys_equals_the_object_level_oracle(tmp_path):
    """the reference walks any selector map (src/predicates.rs:48-53); the device takes 32 label columns per call.  A pod
    with 40 (and 41, and 65) keys is evaluated group by group and the groups' masks ANDed: same masks as the object-level oracle, and the
    reconciler schedules it (first feasible draw of the ANDed row) instead of refusing it."""
    c = synth.make_cluster(P=48, N=90, n_keys=4, n_taints=0, seed=0x40E1)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    for i, n in enumerate(nodes):
        lab = n["metadata"].setdefault("labels", {})
        for k in range(70):
            if (i * 3 + k) % 11:          # most nodes carry most keys ...
                lab[f"wide{k:02d}"] = "v" if (i + k) % 29 else "w"  # ... with a sprinkling of other values
    for i in (5, 6, 7, 20, 33):
        for k in range(70):
            nodes[i]["metadata"]["labels"][f"wide{k:02d}"] = "v"  # a few nodes match everything
    wide = {f"wide{k:02d}": "v" for k in range(40)}
    pods[2]["spec"]["nodeSelector"] = dict(wide)
    pods[11]["spec"]["nodeSelector"] = dict(wide, **{"wide40": "v"})                      # 41 keys
    pods[12]["spec"]["nodeSelector"] = {f"wide{k:02d}": "v" for k in range(65)}           # three groups
    pods[30]["spec"]["nodeSelector"] = dict(wide, **{"wide39": "nobody"})                 # never matches
    pods[31]["spec"]["nodeSelector"] = dict(wide, **(pods[31]["spec"].get("nodeSelector") or {}))  # wide + its ordinary keys
    path = tmp_path / "objs.json"
    write_objects(path, pods, nodes, bound)
    got = tool("masks", path)
    W = (c.N + 63) // 64
    feas, fit = expect_masks(pods, nodes, bound, False, cache=True)
    assert np.array_equal(unhex(got["fit"], W), fit)
    assert np.array_equal(unhex(got["feasible"], W), feas)
    bits = lambda r: int(np.unpackbits(feas[r].view(np.uint8)).sum())  # noqa: E731
    assert 0 < bits(2) <= 5 and 0 < bits(12) <= 5 and bits(30) == 0
    # ... and through the reconciler: outcomes == the oracle's restatement of the batched reference execution, POST by POST
    got = tool("batch", path, 77)
    want, posted = R.reconcile_batch(pods, list(reversed(nodes)), bound, R.SplitMixChooser(77), fail_every=0)  # (the tool's store order is reversed canonical)
    assert [(o["ok"], o["error"], o["bound_to"]) for o in got["outcomes"]] == [(o["ok"], o["error"], o["bound_to"]) for o in want]
    assert [tuple(x) for x in got["posted"]] == posted
    names = [n["metadata"]["name"] for n in nodes]
    for i in (2, 11, 12, 31):
        b = got["outcomes"][i]["bound_to"]
        if b:
            j = names.index(b)
            assert (feas[i, j // 64] >> np.uint64(j % 64)) & np.uint64(1), "a wide pod is bound to a node its ANDed mask allows"
    assert not got["outcomes"][30]["bound_to"]


def small_cluster(seed, P=60, N=12, tight=False):
    c = synth.make_cluster(P=P, N=N, n_keys=4, n_taints=0, seed=seed, binary_suffixes=True)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    pods[3]["spec"]["nodeName"] = nodes[0]["metadata"]["name"]  # an already-bound pod in the batch: skipped (src/main.rs:74-76)
    if tight:  # every node fits about two of these pods: the plain batch over-commits, the sequential one must not
        for n in nodes:
            n["status"]["allocatable"] = {"cpu": "2", "memory": "4Gi"}
        bound = []
        for p in pods:
            for cont in p["spec"]["containers"]:
                if "resources" in cont and cont["resources"].get("requests"):
                    cont["resources"]["requests"] = {"cpu": "400m", "memory": "512Mi"}
            p["spec"].pop("nodeSelector", None)
    return pods, nodes, bound


@pytest.mark.parametrize("seed,fail_every", [(1, 0), (2, 0), (3, 5)])
def test_reconcile_batch_equals_oracle_restatement(tmp_path, seed, fail_every):
    pods, nodes, bound = small_cluster(0xBA7C00 + seed)
    path = tmp_path / "objs.json"
    write_objects(path, pods, nodes, bound)
    got = tool("batch", path, 1000 + seed, fail_every)
    store = list(reversed(nodes))  # the tool hands the store over in reversed canonical order
    want, posted = R.reconcile_batch(pods, store, bound, R.SplitMixChooser(1000 + seed), fail_every=fail_every)
    assert [(o["ok"], o["error"], o["bound_to"]) for o in got["outcomes"]] == [(o["ok"], o["error"], o["bound_to"]) for o in want]
    assert [tuple(x) for x in got["posted"]] == posted
    kinds = {o["error"] for o in want}
    assert None in kinds and "no-node-found" in kinds and (not fail_every or "create-binding-failed" in kinds)
    assert all(o["action"] == ("await_change" if o["ok"] else "requeue_300s") for o in got["outcomes"])  # error_policy, src/main.rs:122-125
    # after the batch its own bindings count against their nodes (the reference re-LISTs per evaluation, src/predicates.rs:34-38)
    by_name = {f"{p['metadata']['namespace']}/{p['metadata']['name']}": p for p in pods}
    state = list(bound)
    for pod_name, node in posted:
        q = json.loads(json.dumps(by_name[pod_name]))
        q["spec"]["nodeName"] = node
        state.append(q)
    for j, node in enumerate(nodes):
        av = R.available_of(node, state)
        assert got["avail_cpu_milli"][j] == av.cpu * 1000 and got["avail_mem_bytes"][j] == av.memory
    assert got["incremental_equals_relist"] is True


def test_reconcile_batch_with_overlapped_posts_equals_oracle_restatement(tmp_path):
    """post_concurrency = 8: the POSTs of the batch overlap, as the reference's concurrent reconciles' do
    (src/main.rs:94-103, 141-144).  Same outcome per pod, the same SET of bindings (their order is the completion order), the same
    snapshot afterwards."""
    pods, nodes, bound = small_cluster(0xBA7C77)
    path = tmp_path / "objs.json"
    write_objects(path, pods, nodes, bound)
    got = tool("batch", path, 1777, 0, 8)
    want, posted = R.reconcile_batch(pods, list(reversed(nodes)), bound, R.SplitMixChooser(1777), fail_every=0)
    assert [(o["ok"], o["error"], o["bound_to"]) for o in got["outcomes"]] == [(o["ok"], o["error"], o["bound_to"]) for o in want]
    assert sorted(tuple(x) for x in got["posted"]) == sorted(posted) and len(posted) > 8
    assert got["incremental_equals_relist"] is True
    by_name = {f"{p['metadata']['namespace']}/{p['metadata']['name']}": p for p in pods}
    state = list(bound)
    for pod_name, node in posted:
        q = json.loads(json.dumps(by_name[pod_name]))
        q["spec"]["nodeName"] = node
        state.append(q)
    for j, node in enumerate(nodes):
        av = R.available_of(node, state)
        assert got["avail_cpu_milli"][j] == av.cpu * 1000 and got["avail_mem_bytes"][j] == av.memory


def test_batching_reconciler_end_to_end_equals_oracle_restatement(tmp_path):
    """PodBatcher (ready_chunks(16)) + run_batches + reconcile_batch through the device: 60 pending pods go out as
    chunks of 16, 16, 16, 12; every chunk is evaluated against the state the earlier chunks left (their bindings count against their
    nodes, as the reference's re-LIST per evaluation would see them, src/predicates.rs:34-38); the draws continue one chooser stream.
    Outcome for outcome, POST for POST and the final snapshot == the oracle's restatement run chunk by chunk."""
    pods, nodes, bound = small_cluster(0x57AEA1, P=60)
    path = tmp_path / "objs.json"
    write_objects(path, pods, nodes, bound)
    got = tool("stream", path, 4242, 16)
    assert (got["batches"], got["largest"]) == (4, 16)
    store, chooser, state = list(reversed(nodes)), R.SplitMixChooser(4242), list(bound)
    by_name = {f"{p['metadata']['namespace']}/{p['metadata']['name']}": p for p in pods}
    want, posted = [], []
    for lo in range(0, 60, 16):
        w, po = R.reconcile_batch(pods[lo:lo + 16], store, state, chooser, fail_every=0)
        want += w
        posted += po
        for pod_name, node in po:
            q = json.loads(json.dumps(by_name[pod_name]))
            q["spec"]["nodeName"] = node
            state.append(q)
    assert [(o["ok"], o["error"], o["bound_to"]) for o in got["outcomes"]] == [(o["ok"], o["error"], o["bound_to"]) for o in want]
    assert [tuple(x) for x in got["posted"]] == posted and len(posted) > 10
    for j, node in enumerate(nodes):
        av = R.available_of(node, state)
        assert got["avail_cpu_milli"][j] == av.cpu * 1000 and got["avail_mem_bytes"][j] == av.memory


@pytest.mark.parametrize("seed,fail_every,tight", [(1, 0, True), (2, 4, True), (3, 0, False)])
def test_reconcile_batch_sequential_equals_oracle_restatement(tmp_path, seed, fail_every, tight):
    pods, nodes, bound = small_cluster(0x5E0000 + seed, P=48, N=10, tight=tight)
    path = tmp_path / "objs.json"
    write_objects(path, pods, nodes, bound)
    got = tool("sequential", path, 2000 + seed, fail_every)
    store = list(reversed(nodes))
    want, posted, rounds, conflicts, state = R.reconcile_batch_sequential(pods, store, bound, R.SplitMixChooser(2000 + seed), fail_every=fail_every)
    assert [(o["ok"], o["error"], o["bound_to"]) for o in got["outcomes"]] == [(o["ok"], o["error"], o["bound_to"]) for o in want]
    assert [tuple(x) for x in got["posted"]] == posted
    assert (got["rounds"], got["conflicts"]) == (rounds, conflicts)
    if tight:
        assert rounds > 1 and conflicts > 0, "the case must exercise the deferral"
    assert got["incremental_equals_relist"] is True
    # the snapshot the device was left with == allocatable - LIST over the final state (exact), and nothing is over-committed
    for j, node in enumerate(nodes):
        av = R.available_of(node, state)
        assert got["avail_cpu_milli"][j] == av.cpu * 1000 and got["avail_mem_bytes"][j] == av.memory
        if tight:
            assert av.cpu >= 0 and av.memory >= 0
