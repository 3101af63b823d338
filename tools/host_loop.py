#!/usr/bin/env python
"""Throughput of the callers around the kernel through the C++ host mirror, on the GPU box: JSON objects
-> host encoder -> device evaluation + pick -> binding POSTs (recorded) -> snapshot update.  `batch` = reconcile_batch (every pod
against one snapshot, the reference's racing semantics); `sequential` = reconcile_batch_sequential (rounds, no over-commit).
The numbers include the host's string parsing and the per-round device calls; the snapshot build (one LIST per node through the
test double + encode + ksched_set_nodes) is done before the clock starts.

    python tools/host_loop.py [--sizes 5000x500,20000x2000] [--modes batch,sequential] [--reps 3] [--post-concurrency 1] [--warn]
`--sizes 100000x5000` is the C3-size batch bench.py's `end_to_end.objects` quotes.  Without --warn the WARN level is off (OBJECTS_EVAL_QUIET
silences it: the reference's warn!() lines go to stderr by the thousand otherwise and are what the run then measures)."""
import argparse, json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_scheduler_rs_reference_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.environ.get("OBJECTS_EVAL_BIN") or os.path.join(ROOT, "tests", "cpp", "objects_eval")  # (OBJECTS_EVAL_BIN: a sanitizer build, tools/sanitize.sh)


def objects_file(P, N):
    c = synth.make_cluster(P=P, N=N, n_keys=8, n_taints=0, seed=0x100 + P)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"name": "loop", "pods": c.pod_objects(), "nodes": c.node_objects(), "bound": c.bound_pod_objects(), "samples": []}, f)
        return f.name


def run(mode, path, reps=1, post_concurrency=1, warn=False, timeout=900):
    """-> (the tool's JSON, its stderr lines that start with "reconcile_batch" (KSCHED_HOST_TIMING), process wall seconds)"""
    env = dict(os.environ, OBJECTS_EVAL_QUIET="1", KSCHED_HOST_TIMING=os.environ.get("KSCHED_HOST_TIMING", "1"), OBJECTS_EVAL_REPS=str(reps))
    if warn:
        env["OBJECTS_EVAL_WARN"] = "1"
    t0 = time.perf_counter()
    r = subprocess.run([TOOL, mode, path, "4242", "0", str(post_concurrency)], capture_output=True, text=True, env=env, timeout=timeout)
    wall = time.perf_counter() - t0
    if "Sanitizer" in r.stderr or "runtime error:" in r.stderr:  # a sanitizer build (OBJECTS_EVAL_BIN): its reports are the point
        sys.stdout.write(r.stderr)
    if r.returncode:
        raise RuntimeError(f"objects_eval {mode} failed: {r.stderr[-400:]}")
    # (RCCL prints a five-line version banner to stdout when a communicator is created -- KSCHED_SHARDED=1 --, in the middle of the tool's JSON)
    banner = ("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl path")
    kept = []
    for ln in r.stdout.splitlines(keepends=True):
        if ln.startswith("{") and ln[1:].startswith(banner):
            kept.append("{")
        elif not ln.startswith(banner):
            kept.append(ln)
    return json.loads("".join(kept)), [ln for ln in r.stderr.splitlines() if ln.startswith(("reconcile_batch", "  phase"))], wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="5000x500,20000x2000")
    ap.add_argument("--modes", default="batch,sequential")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--post-concurrency", type=int, default=1)
    ap.add_argument("--warn", action="store_true")
    a = ap.parse_args()
    print(f"# host cores: {os.cpu_count()}")
    for size in a.sizes.split(","):
        P, N = (int(x) for x in size.split("x"))
        path = objects_file(P, N)
        for mode in a.modes.split(","):
            try:
                d, timing, wall = run(mode, path, a.reps, a.post_concurrency, a.warn)
            except Exception as e:  # noqa: BLE001
                print(mode, P, N, "FAILED", e)
                continue
            secs = d.get("seconds_all") or [d["seconds"]]
            best = min(secs)
            print(f"{mode:10s} {P} pods x {N} nodes: {d['posted_count']} bound in {best * 1e3:.1f} ms (best of {len(secs)}: " + " ".join(f"{s * 1e3:.1f}" for s in secs) +
                  f") = {d['posted_count'] / best:.0f} pods/s" + (f" ({d['rounds']} rounds, {d['conflicts']} deferrals)" if mode == "sequential" else "") +
                  f"   [process wall {wall:.1f} s incl. JSON load + snapshot]")
            per = len(timing) // max(1, len(secs) + (1 if any(ln.startswith("  phase") for ln in timing) else 0)) if any(ln.startswith("  phase") for ln in timing) else 1
            for ln in (timing[-per:] if per > 1 else timing[-len(secs):]):  # (phases: the last repeat's lines)
                print("    " + ln)
        os.unlink(path)


if __name__ == "__main__":
    main()
