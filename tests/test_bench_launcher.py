"""bench.py --gpus N started plainly (no WORLD_SIZE in the environment) becomes its own launcher: one process per GPU under
torch.distributed.run on this node, rendezvous on 127.0.0.1 (`python bench.py --gpus 8` used to exit with a usage message, which would have
ended the one chance at a scaling curve).  No GPU needed: the launcher is replaced by a recorder."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_world_size_relaunches_itself_under_torch_distributed_run(tmp_path):
    # a stand-in interpreter: records its argv and exits 0 (what os.execv hands over to)
    rec = tmp_path / "argv.json"
    fake_torch = tmp_path / "fake_site"
    (fake_torch / "torch" / "distributed").mkdir(parents=True)
    (fake_torch / "torch" / "__init__.py").write_text("")
    (fake_torch / "torch" / "distributed" / "__init__.py").write_text("")
    (fake_torch / "torch" / "distributed" / "run.py").write_text(
        "import json, os, sys\njson.dump({'argv': sys.argv[1:], 'world': os.environ.get('WORLD_SIZE')}, open(os.environ['REC'], 'w'))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["REC"] = str(rec)
    # the re-launched `python -m torch.distributed.run` resolves torch from PYTHONPATH first: the recorder; bench.py itself imports the real one
    # only after the launch decision, so the recorder is what it finds too -- which is fine, the decision needs no torch
    env["PYTHONPATH"] = str(fake_torch)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "re-launching" in r.stderr
    got = json.load(open(rec))
    a = got["argv"]
    assert "--nnodes=1" in a and "--nproc-per-node=8" in a and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(a[a.index("--master-port") + 1]) < 65536
    i = a.index(os.path.join(ROOT, "bench.py"))
    assert a[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"], "the same command line goes to every rank"


def test_world_size_that_contradicts_gpus_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2 != --gpus 4" in (r.stderr + r.stdout)


def test_a_rank_generates_only_its_rows_of_every_input_batch():
    """bench.py's input rotation: rank r's rows of batch b are pods [b * P_total + lo_r, b * P_total + hi_r) of the seed's pod sequence.  The generator's
    streams are counter-based, so a window of pods (synth pod_offset) equals the same rows of a cluster generated whole: the ranks' pieces tile the
    batches exactly, and batch 0 at N = 1 is the workload's standard batch."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from kube_scheduler_rs_reference_amd import synth
    for cfg in ("C3", "C5", "C3h"):
        whole = synth.make_config(cfg, P=3 * 400, N=257)          # three batches of 400 pods
        for b in range(3):
            for lo, hi in ((0, 134), (134, 268), (268, 400)):     # three ranks' shards of a batch
                part = synth.make_config(cfg, P=hi - lo, N=257, pod_offset=b * 400 + lo)
                rows = slice(b * 400 + lo, b * 400 + hi)
                for f in ("req_cpu", "req_mem", "pod_tol", "samples", "pod_ncont", "pod_has_req", "cont_cpu", "cont_mem"):
                    assert np.array_equal(getattr(whole, f)[rows], getattr(part, f)), (cfg, b, lo, f)
                assert np.array_equal(whole.pod_sel[:, rows], part.pod_sel)
                assert np.array_equal(whole.avail_cpu, part.avail_cpu) and np.array_equal(whole.avail_mem, part.avail_mem)
                assert np.array_equal(whole.node_labels, part.node_labels) and np.array_equal(whole.node_taints, part.node_taints)
        std = synth.make_config(cfg, P=400, N=257)
        assert np.array_equal(std.req_cpu, whole.req_cpu[:400]) and np.array_equal(std.samples, whole.samples[:400])


FAKE_ROCPROF = r'''#!/usr/bin/env python3
# stand-in for rocprofv3 (no GPU here): records the command it was asked to profile and writes the counter CSV rocprofv3 would write
import json, os, sys
a = sys.argv[1:]
out = a[a.index("-d") + 1]
counter = a[a.index("--pmc") + 1]
cmd = a[a.index("--") + 1:]
assert "--kernel-trace" in a and "--sys-trace" not in a and "-s" not in a, "counters are collected with --kernel-trace only"
with open(os.environ["FAKE_ROCPROF_LOG"], "a") as f:
    f.write(json.dumps({"counter": counter, "cmd": cmd, "cwd": os.getcwd()}) + "\n")
if os.environ.get("FAKE_ROCPROF_FAIL") == counter:
    sys.exit(7)
os.makedirs(os.path.join(out, "host", "1234"), exist_ok=True)
rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
if cmd[0].endswith("pmc_calib"):
    vals = {"FETCH_SIZE": {"calib_read_flat16(float4 const*, float4*, unsigned long)": 262144.0, "calib_write_flat16(float4*, unsigned long)": 10.0},
            "WRITE_SIZE": {"calib_write_tile128(unsigned long*, unsigned int)": 156250.0, "calib_write_flat16(float4*, unsigned long)": 524288.0}}[counter]
    for i, (k, v) in enumerate(vals.items()):
        rows.append('%d,"%s",%s,%f' % (i + 1, k, counter, v))
else:
    per = {"FETCH_SIZE": 5795.0, "WRITE_SIZE": 63707.0}[counter]
    for d in range(1, 4):  # three dispatches, the counter split over two rows each (one per XCD group: per_kernel sums the rows of a dispatch)
        for half in (0.25, 0.75):
            rows.append('%d,"void ksched::k_eval_fused<true, true, false, false, false, 2>(unsigned long const*, ksched::FusedArgs)",%s,%f' % (d, counter, per * half))
    rows.append('9,"ksched::k_sort_runs(ksched::SortArgs)",%s,1.0' % counter)
open(os.path.join(out, "host", "1234", "p_counter_collection.csv"), "w").write("\n".join(rows) + "\n")
'''


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # (top level only: arguments are parsed and torch imported inside main())
    return mod


def test_live_traffic_runs_four_calibrated_counter_passes_and_falls_back_when_one_fails(tmp_path, monkeypatch):
    """bench.py --live-traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (--pmc with --kernel-trace only), each over
    tools/pmc_calib (known byte counts) and over bench.py itself with 10 steps; bytes per launch = counter x calibrated bytes per count, summed over the
    rows of a dispatch, averaged over the mask kernel's dispatches.  No GPU here: rocprofv3 is a stand-in that writes the CSV the real one writes."""
    calib = os.path.join(ROOT, "tools", "pmc_calib")
    if not os.path.exists(calib):
        import pytest
        pytest.skip("tools/pmc_calib not built (make tools)")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    fake = bindir / "rocprofv3"
    fake.write_text(FAKE_ROCPROF)
    fake.chmod(0o755)
    log = tmp_path / "calls.jsonl"
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("FAKE_ROCPROF_LOG", str(log))
    bench = _bench_module()
    rec, why = bench.live_traffic("C3", "auto", ["--fused-pick", "1", "--input-batches", "6"])
    assert why is None and rec["kernel"] == "k_eval_fused" and rec["dispatches"] == [3, 3]
    assert rec["fetch_bytes_per_count"] == (512 << 20) / 262144.0 == 2048.0 and rec["write_bytes_per_count"] == 125000 * 10 * 128 / 156250.0 == 1024.0
    assert rec["fetch_bytes"] == 5795.0 * 2048.0 and rec["write_bytes"] == 63707.0 * 1024.0
    assert rec["hbm_bytes_per_launch"] == rec["fetch_bytes"] + rec["write_bytes"]
    calls = [json.loads(ln) for ln in open(log)]
    assert [c["counter"] for c in calls] == ["FETCH_SIZE", "FETCH_SIZE", "WRITE_SIZE", "WRITE_SIZE"], "one counter per pass"
    assert calls[0]["cmd"] == [calib] and calls[2]["cmd"] == [calib]
    child = calls[1]["cmd"]
    assert child[1] == os.path.join(ROOT, "bench.py") and child[child.index("--workload") + 1] == "C3" and child[child.index("--steps") + 1] == "10"
    assert child[child.index("--live-traffic") + 1] == "off", "the passes must not start passes of their own"
    assert "--no-cpu-baseline" in child and child[-4:] == ["--fused-pick", "1", "--input-batches", "6"]
    assert all(c["cwd"] == "/tmp" for c in calls)
    # a pass that fails: no figure, a reason -- the line then keeps the committed figure and says why (bench.py main)
    monkeypatch.setenv("FAKE_ROCPROF_FAIL", "WRITE_SIZE")
    rec, why = bench.live_traffic("C3", "auto", [])
    assert rec is None and "calib_write" in why and "exited 7" in why
