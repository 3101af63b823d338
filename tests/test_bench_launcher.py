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
