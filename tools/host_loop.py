#!/usr/bin/env python
"""Throughput of the callers around the kernel through the C++ host mirror, on the GPU box: JSON objects
-> host encoder -> device evaluation + pick -> binding POSTs (recorded) -> snapshot update.  `batch` = reconcile_batch (every pod
against one snapshot, the reference's racing semantics); `sequential` = reconcile_batch_sequential (rounds, no over-commit).
The numbers include the host's string parsing and the per-round device calls; the snapshot build (one LIST per node through the
test double + encode + ksched_set_nodes) is done before the clock starts."""
import json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_scheduler_rs_reference_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "cpp", "objects_eval")
for P, N in ((5_000, 500), (20_000, 2_000)):
    c = synth.make_cluster(P=P, N=N, n_keys=8, n_taints=0, seed=0x100 + P)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"name": "loop", "pods": c.pod_objects(), "nodes": c.node_objects(), "bound": c.bound_pod_objects(), "samples": []}, f)
        path = f.name
    for mode in ("batch", "sequential"):
        t0 = time.perf_counter()
        r = subprocess.run([TOOL, mode, path, "4242"], capture_output=True, text=True, env=dict(os.environ, OBJECTS_EVAL_QUIET="1"), timeout=900)
        wall = time.perf_counter() - t0
        if r.returncode:
            print(mode, P, N, "FAILED", r.stderr[-300:])
            continue
        # (RCCL prints a five-line version banner to stdout when a communicator is created -- KSCHED_SHARDED=1 --, in the middle of the tool's JSON)
        banner = ("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl path")
        kept = []
        for ln in r.stdout.splitlines(keepends=True):
            if ln.startswith("{") and ln[1:].startswith(banner):
                kept.append("{")
            elif not ln.startswith(banner):
                kept.append(ln)
        d = json.loads("".join(kept))
        print(f"{mode:10s} {P} pods x {N} nodes: {d['posted_count']} bound in {d['seconds'] * 1e3:.1f} ms = {d['posted_count'] / d['seconds']:.0f} pods/s"
              + (f" ({d['rounds']} rounds, {d['conflicts']} deferrals)" if mode == "sequential" else "") + f"   [process wall {wall:.1f} s incl. JSON load + snapshot]")
    os.unlink(path)
