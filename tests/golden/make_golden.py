"""Generates tests/golden/*.npz -- committed golden input/output vectors for the hot path.

The reference is Rust and cannot be built or imported here (no cargo/rustc),
so these are NOT outputs of the reference binary.  They are produced by the exact, object-level
Python restatement (oracle/oracle_ref.py: Fractions on Kubernetes-shaped dicts, one per-pair
evaluation at a time) on seeded synthetic clusters, and stored next to the encoded columns of the
same clusters.  They pin (a) the oracle's C restatements and (b) the HIP path against drift.

Next to every <name>.npz the same cluster is exported as Kubernetes JSON objects, <name>_objects.json
({"pods", "nodes", "bound", "samples"}): the input a maintainer with cargo feeds to the REFERENCE's own predicates
(rust/src/predicates/parity_dump.rs, one command: rust/pin_parity.sh) to produce tests/golden/ref_<name>.json, which
tests/test_reference_fixtures.py then compares with these fixtures -- that is what pins resource-fit to the reference.

Two further object-only cases have no encoded columns:
  binsuffix_60x40   memory spelled with Ki / Mi (believed exact in kube_quantity 0.6.1)
  subunit_22x8      sub-milli CPU / sub-byte memory at exact-fit boundaries: OUTSIDE D; the snapshot's column unit follows
  hazard_gi_24x10   Gi / Ti / exponent / fractional spellings at exact-fit boundaries: OUTSIDE the parity domain D; the
                    expected masks follow true Kubernetes semantics (exact powers of 1024).  A reference run that
                    differs here documents kube_quantity's suspected f32 scale conversion, it does not fail parity.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from kube_scheduler_rs_reference_amd import pack_mask, synth  # noqa: E402
from oracle import oracle_ref as R  # noqa: E402

CASES = {
    # BASELINE.json configs[0]: 100 pods x 20 nodes
    "c1_100x20": dict(P=100, N=20, n_keys=8, n_taints=0, seed=0x5EED0000),
    # ragged: N not a multiple of 64, crosses two words; taints on
    "ragged_70x130_taints": dict(P=70, N=130, n_keys=8, n_taints=16, seed=0x5EED0101),
    # a single node, a single word
    "one_node_33x1": dict(P=33, N=1, n_keys=8, n_taints=3, seed=0x5EED0202),
}


def hex_rows(mask):
    return [[f"{int(w):016x}" for w in row] for row in mask]


def dump_objects(name, pods, nodes, bound, samples, domain):
    doc = {"name": name, "domain": domain, "p": len(pods), "n": len(nodes), "attempts": R.ATTEMPTS,
           "note": "nodes are in canonical order (ascending metadata.name); mask bit n of a pod row = node n",
           "pods": pods, "nodes": nodes, "bound": bound, "samples": [[int(x) for x in r] for r in samples]}
    with open(os.path.join(HERE, name + "_objects.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")


def dump_expected(name, pods, nodes, bound, samples, use_taint):
    """Object-only cases: expected masks as hex words (what ref_<name>.json holds for the reference side)."""
    P, N = len(pods), len(nodes)
    feas, fit = R.eval_matrix(pods, nodes, bound, use_taint=use_taint)
    sel, _ = R.eval_matrix(pods, nodes, bound, use_fit=False)
    pk = lambda m: pack_mask(np.array(m, dtype=bool).reshape(P, N))  # noqa: E731
    picks = [(-1 if (b := R.select_node_for_pod(p, nodes, bound, [int(s) for s in samples[i]])) is None else b) for i, p in enumerate(pods)]
    doc = {"name": name, "p": P, "n": N, "fit": hex_rows(pk(fit)), "sel": hex_rows(pk(sel)),
           "feasible_fit_and_sel": hex_rows(pk(fit) & pk(sel)), "sampled": picks,
           "source": "oracle/oracle_ref.py (exact Fractions, true Kubernetes quantity semantics)"}
    with open(os.path.join(HERE, name + "_expected.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")


def hazard_case():
    """Hand-built: quantities OUTSIDE the canonical domain D at exact-fit boundaries."""
    GI = 1 << 30
    allocs = [("4", "1Gi"), ("8", "8Gi"), ("7500m", "16Gi"), ("0.5", "1.5Gi"), ("2", "1Ti"), ("16", "64Gi"),
              ("1e1", "1e10"), ("32", "34359738368"), ("3", "3G"), ("64", "0.5Ti")]
    nodes = [{"metadata": {"name": f"hz-node-{i:02d}", "labels": {"zone": "a" if i % 2 else "b"}},
              "status": {"allocatable": {"cpu": c, "memory": m}}} for i, (c, m) in enumerate(allocs)]
    reqs = [("1", str(GI)), ("1", str(GI + 1)), ("1", "1Gi"), ("1", "1024Mi"), ("1", "1048576Ki"), ("500m", "1.5Gi"),
            ("0.5", "1610612736"), ("0.5", "1610612737"), ("2", "8Gi"), ("2", str(8 * GI)), ("2", str(8 * GI + 1)),
            ("7500m", "16Gi"), ("7501m", "16Gi"), ("7.5", str(16 * GI)), ("1", "1Ti"), ("1", str(1 << 40)), ("1", str((1 << 40) + 1)),
            ("10", "1e10"), ("10", "10G"), ("10", "10000000001"), ("3", "3G"), ("3", "3000000000"), ("3", "2.5Gi"), ("1", "512Gi")]
    pods = [{"metadata": {"name": f"hz-pod-{i:02d}", "namespace": "hz"},
             "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": c, "memory": m}}}]}}
            for i, (c, m) in enumerate(reqs)]
    pods[3]["spec"]["nodeSelector"] = {"zone": "a"}
    bound = [{"metadata": {"name": "hz-bound-0", "namespace": "hz"},
              "spec": {"nodeName": "hz-node-05", "containers": [{"name": "c", "resources": {"requests": {"cpu": "1", "memory": "32Gi"}}}]}}]
    samples = np.array([[(i * 7 + t * 3) % len(nodes) for t in range(R.ATTEMPTS)] for i in range(len(pods))], dtype=np.uint32)
    return pods, nodes, bound, samples


def subunit_case():
    """Hand-built: quantities FINER than a milli-core / a byte -- cpu "100u" / "1500n" / "0.0005", memory "100m" (a tenth of a byte) / "1n" --
    at exact-fit boundaries.  The reference parses any quantity (src/util.rs:64-69, src/predicates.rs:29-31) and compares decimals; the
    product picks the snapshot's column unit from the finest value present (host/encoder.hpp cpu_unit_nanos) and must give the same bits."""
    GI = 1 << 30
    allocs = [("2", "1000"), ("1500n", "1"), ("0.0005", "0.5"), ("4", "4Gi"), ("250u", "1k"), ("1", "1500m"), ("1000001n", "1"), ("8", "8Gi")]
    nodes = [{"metadata": {"name": f"su-node-{i:02d}", "labels": {"zone": "a" if i % 2 else "b"}},
              "status": {"allocatable": {"cpu": c, "memory": m}}} for i, (c, m) in enumerate(allocs)]
    reqs = [("1999900u", "999900m"), ("1999900001n", "1"), ("1", "999901m"), ("1500n", "1"), ("1501n", "1"), ("500u", "0.5"), ("500u", "500m"),
            ("501u", "0.5"), ("0.5m", "1"), ("250u", "1k"), ("250001n", "1"), ("1", "1500m"), ("1", "1501m"), ("1000001n", "1"), ("1000002n", "1"),
            ("7999999999n", str(8 * GI - 1)), ("8", "8Gi"), ("0", "0"), ("100u", "100m"), ("4", "4Gi"), ("3999999999n", "4294967295999999999n"), ("1n", "1n")]
    pods = [{"metadata": {"name": f"su-pod-{i:02d}", "namespace": "su"},
             "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": c, "memory": m}}}]}}
            for i, (c, m) in enumerate(reqs)]
    pods[5]["spec"]["containers"].append({"name": "half", "resources": {"requests": {"cpu": "0", "memory": "0"}}})
    pods[18]["spec"]["nodeSelector"] = {"zone": "b"}
    bound = [{"metadata": {"name": "su-bound-0", "namespace": "su"},
              "spec": {"nodeName": "su-node-00", "containers": [{"name": "c", "resources": {"requests": {"cpu": "100u", "memory": "100m"}}}]}},
             {"metadata": {"name": "su-bound-1", "namespace": "su"},
              "spec": {"nodeName": "su-node-07", "containers": [{"name": "c", "resources": {"requests": {"cpu": "1n", "memory": "1n"}}}]}}]
    samples = np.array([[(i * 3 + t * 5) % len(nodes) for t in range(R.ATTEMPTS)] for i in range(len(pods))], dtype=np.uint32)
    return pods, nodes, bound, samples


def typical_case():
    """Hand-built: the spellings real manifests and kubelets use -- cpu "500m" / "2" / "0.5", memory "128Mi" / "1Gi" / "512Mi",
    allocatable "32779148Ki" / "7910m" -- including pods that fit a node EXACTLY (where a reading that inflates Gi by 1.6e-7 shows)."""
    allocs = [("8", "32779148Ki"), ("7910m", "32779148Ki"), ("4", "16Gi"), ("3920m", "15031Mi"), ("16", "64Gi"), ("2", "4Gi"),
              ("96", "196608Mi"), ("1", "2Gi"), ("32", "128Gi"), ("8", "31Gi"), ("4", "8G"), ("64", "256Gi")]
    nodes = [{"metadata": {"name": f"ty-node-{i:02d}", "labels": {"pool": "gp" if i % 3 else "hm", "kubernetes.io/os": "linux"}},
              "status": {"allocatable": {"cpu": c, "memory": m}}} for i, (c, m) in enumerate(allocs)]
    reqs = [("500m", "1Gi"), ("500m", "512Mi"), ("250m", "128Mi"), ("100m", "64Mi"), ("1", "2Gi"), ("2", "4Gi"), ("4", "16Gi"), ("4", "8G"),
            ("0.5", "1Gi"), ("1500m", "3Gi"), ("8", "32779148Ki"), ("7910m", "32779148Ki"), ("16", "64Gi"), ("3920m", "15031Mi"),
            ("32", "128Gi"), ("1", "2147483648"), ("2", "4294967296"), ("1", "2147483649"), ("64", "256Gi"), ("8", "31Gi"),
            ("10m", "16Mi"), ("50m", "100M"), ("2", "1500Mi"), ("1", "1G"), ("3", "6Gi"), ("6", "24Gi"), ("12", "48Gi"), ("200m", "256Mi"),
            ("1", "1.5Gi"), ("750m", "768Mi"), ("4", "17179869184"), ("4", "17179869185"), ("16", "68719476736"), ("16", "68719476737"),
            ("96", "196608Mi"), ("96", "192Gi"), ("2", "4000Mi"), ("1", "1025Mi"), ("8", "16Gi"), ("4", "7Gi")]
    pods = [{"metadata": {"name": f"ty-pod-{i:02d}", "namespace": "ty"},
             "spec": {"containers": [{"name": "app", "resources": {"requests": {"cpu": c, "memory": m}}}]}}
            for i, (c, m) in enumerate(reqs)]
    pods[5]["spec"]["containers"].append({"name": "sidecar", "resources": {"requests": {"cpu": "100m", "memory": "128Mi"}}})
    pods[9]["spec"]["nodeSelector"] = {"pool": "hm"}
    pods[20]["spec"]["nodeSelector"] = {"kubernetes.io/os": "linux"}
    bound = [{"metadata": {"name": "ty-bound-0", "namespace": "ty"},
              "spec": {"nodeName": "ty-node-04", "containers": [{"name": "c", "resources": {"requests": {"cpu": "4", "memory": "16Gi"}}}]}},
             {"metadata": {"name": "ty-bound-1", "namespace": "ty"},
              "spec": {"nodeName": "ty-node-00", "containers": [{"name": "c", "resources": {"requests": {"cpu": "1500m", "memory": "6Gi"}}}]}},
             {"metadata": {"name": "ty-bound-2", "namespace": "ty"},
              "spec": {"nodeName": "ty-node-08", "containers": [{"name": "c", "resources": {"requests": {"cpu": "30", "memory": "120Gi"}}}]}}]
    samples = np.array([[(i * 5 + t * 7) % len(nodes) for t in range(R.ATTEMPTS)] for i in range(len(pods))], dtype=np.uint32)
    return pods, nodes, bound, samples


SPELLINGS = ["0", "1", "2", "500m", "250m", "100m", "10m", "1500m", "7910m", "0.5", "1.5", "64Mi", "128Mi", "512Mi", "768Mi", "1025Mi", "15031Mi", "196608Mi",
             "32779148Ki", "1Gi", "2Gi", "4Gi", "16Gi", "31Gi", "64Gi", "256Gi", "1.5Gi", "1Ti", "0.5Ti", "1Pi", "1G", "8G", "100M", "1k", "1T", "10E",
             "1073741824", "17179869184", "1n", "100u", "1e3", "1E3", "129e6", "5.", "+5", "-5m"]


def wide_case():
    """Selector maps wider than the device's 32 label columns per call (src/predicates.rs:48-53 walks any map): pods with 40, 41 and 65
    keys, one that can never match, one with wide + ordinary keys.  Quantities stay in domain D."""
    c = synth.make_cluster(P=48, N=90, n_keys=4, n_taints=0, seed=0x40E1)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    for i, n in enumerate(nodes):
        lab = n["metadata"].setdefault("labels", {})
        for k in range(70):
            if (i * 3 + k) % 11:
                lab[f"wide{k:02d}"] = "v" if (i + k) % 29 else "w"
    for i in (5, 6, 7, 20, 33):
        for k in range(70):
            nodes[i]["metadata"]["labels"][f"wide{k:02d}"] = "v"
    wide = {f"wide{k:02d}": "v" for k in range(40)}
    pods[2]["spec"]["nodeSelector"] = dict(wide)
    pods[11]["spec"]["nodeSelector"] = dict(wide, **{"wide40": "v"})
    pods[12]["spec"]["nodeSelector"] = {f"wide{k:02d}": "v" for k in range(65)}
    pods[30]["spec"]["nodeSelector"] = dict(wide, **{"wide39": "nobody"})
    pods[31]["spec"]["nodeSelector"] = dict(wide, **(pods[31]["spec"].get("nodeSelector") or {}))
    return pods, nodes, bound, c.samples


def dump_readings(name, pods, nodes, bound):
    """Both expectations of the fit mask: exact Kubernetes semantics (what the product implements) and kube_quantity 0.6.1 as
    recalled (oracle_ref.KubeQuantity061).  ref_<name>.json of a real reference run is compared with both (tests/test_quantity_readings.py)."""
    P, N = len(pods), len(nodes)
    _, fit = R.eval_matrix(pods, nodes, bound)
    kq = R.kq061_fit_matrix(pods, nodes, bound)
    exact_bits = np.array(fit, dtype=bool).reshape(P, N)
    kq_bits = np.array([bool(x) for x in kq], dtype=bool).reshape(P, N)  # (a pair the recalled parser would panic on counts as infeasible: parity_dump.rs does the same)
    panics = [[i // N, i % N] for i, x in enumerate(kq) if x is None]
    differ = [[int(p), int(n)] for p, n in np.argwhere(exact_bits != kq_bits)]
    doc = {"name": name, "p": P, "n": N, "fit_exact_kubernetes": hex_rows(pack_mask(exact_bits)), "fit_kube_quantity_0_6_1_as_recalled": hex_rows(pack_mask(kq_bits)),
           "pairs_where_the_readings_differ": differ, "pairs_the_recalled_parser_rejects": panics,
           "source": "oracle/oracle_ref.py: eval_matrix (exact Fractions) and kq061_fit_matrix (Decimal + f32 scale factors, 7 significant digits)"}
    with open(os.path.join(HERE, name + "_readings.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")
    return len(differ), len(panics)


def dump_spellings():
    rows = []
    for s in SPELLINGS:
        try:
            a = R.parse_quantity(s)
            exact = f"{a.numerator}/{a.denominator}" if a.denominator != 1 else str(a.numerator)
        except R.ReferencePanic:
            exact = "rejected"
        try:
            b = R.KubeQuantity061.parse(s).in_units()
            kq = f"{b.numerator}/{b.denominator}" if b.denominator != 1 else str(b.numerator)
        except R.ReferencePanic:
            kq = "rejected"
        rows.append({"spelling": s, "exact_kubernetes": exact, "kube_quantity_0_6_1_as_recalled": kq, "agree": exact == kq})
    with open(os.path.join(HERE, "quantity_readings.json"), "w") as f:
        json.dump({"note": "value of the spelling as a request entering the accumulator seeded \"0\" (src/util.rs:25-26,65,68), in cores / bytes",
                   "rows": rows}, f, indent=1)
        f.write("\n")


def main():
    # object-only cases
    c = synth.make_cluster(P=60, N=40, n_keys=8, n_taints=0, seed=0x5EED0303, binary_suffixes=True)
    pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
    dump_objects("binsuffix_60x40", pods, nodes, bound, c.samples, "D + Ki/Mi memory spellings (believed exact)")
    dump_expected("binsuffix_60x40", pods, nodes, bound, c.samples, use_taint=False)
    pods, nodes, bound, samples = hazard_case()
    dump_objects("hazard_gi_24x10", pods, nodes, bound, samples, "OUTSIDE D: Gi/Ti/exponent/fractional spellings (hazard list, SURVEY.md 8c)")
    dump_expected("hazard_gi_24x10", pods, nodes, bound, samples, use_taint=False)
    print("hazard_gi_24x10: readings differ on %d pairs, the recalled parser rejects %d" % dump_readings("hazard_gi_24x10", pods, nodes, bound))
    pods, nodes, bound, samples = typical_case()
    dump_objects("typical_specs_40x12", pods, nodes, bound, samples, "typical real spellings (500m / 1Gi / 512Mi / 32779148Ki): Gi and above are OUTSIDE D")
    dump_expected("typical_specs_40x12", pods, nodes, bound, samples, use_taint=False)
    print("typical_specs_40x12: readings differ on %d pairs, the recalled parser rejects %d" % dump_readings("typical_specs_40x12", pods, nodes, bound))
    pods, nodes, bound, samples = subunit_case()
    dump_objects("subunit_22x8", pods, nodes, bound, samples, "OUTSIDE D: sub-milli CPU (100u, 1500n) and sub-byte memory (100m, 1n) at exact-fit boundaries")
    dump_expected("subunit_22x8", pods, nodes, bound, samples, use_taint=False)
    pods, nodes, bound, samples = wide_case()
    dump_objects("wide_selectors_48x90", pods, nodes, bound, samples, "D, selector maps of 40 / 41 / 65 keys (more than the device's 32 label columns per call)")
    dump_expected("wide_selectors_48x90", pods, nodes, bound, samples, use_taint=False)
    dump_spellings()
    for name, kw in CASES.items():
        c = synth.make_cluster(**kw)
        pods, nodes, bound = c.pod_objects(), c.node_objects(), c.bound_pod_objects()
        dump_objects(name, pods, nodes, bound, c.samples, "D (SURVEY.md section 8c): cpu '<n>' / '<n>m', memory plain integer bytes")
        use_taint = kw["n_taints"] > 0
        feas, fit = R.eval_matrix(pods, nodes, bound, use_taint=use_taint)
        feas = pack_mask(np.array(feas, dtype=bool).reshape(c.P, c.N))
        fit = pack_mask(np.array(fit, dtype=bool).reshape(c.P, c.N))
        sampled = np.array([(-1 if (b := R.select_node_for_pod(p, nodes, bound, [int(s) for s in c.samples[i]])) is None else b)
                            for i, p in enumerate(pods)], dtype=np.int32) if not use_taint else None
        bestfit = np.array([(-1 if (b := R.pick_bestfit(p, nodes, bound, use_taint=use_taint)) is None else b)
                            for p in pods], dtype=np.int32)
        out = dict(avail_cpu=c.avail_cpu, avail_mem=c.avail_mem, node_labels=c.node_labels, node_taints=c.node_taints,
                   req_cpu=c.req_cpu, req_mem=c.req_mem, pod_sel=c.pod_sel, pod_tol=c.pod_tol, samples=c.samples,
                   feasible=feas, fit=fit, bestfit=bestfit, n_taints=np.int64(kw["n_taints"]))
        if sampled is not None:
            out["sampled"] = sampled
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "density", np.unpackbits(feas.view(np.uint8)).sum() / (c.P * c.N))


if __name__ == "__main__":
    main()
