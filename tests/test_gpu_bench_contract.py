"""bench.py's output contract (the driver parses this line): ONE JSON line, the metric / unit / workload names of BASELINE.json,
the `roofline` and `cpu_baseline` objects, value consistent with steps and ms_per_step.  Short run (reduced steps, small CPU sample)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_default_line_contract(built):
    d = run_bench("--steps", "100", "--warmup", "5", "--ramp-ms", "10", "--kernel-samples", "16")
    assert d["metric"] == "pod x node predicate evals/s" and d["unit"] == "evals/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 100 and d["warmup"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "int64" and d["data"] == "synthetic"
    c = d["config"]
    assert c["workload"].startswith("C3: 100k pods x 5k nodes") and c["pods_total"] == 100_000 and c["nodes"] == 5_000 and c["kernel"] == "fused"
    assert c["mask_written"] is True and 0.2 < c["bound_fraction"] < 0.8
    assert abs(d["value"] - 100_000 * 5_000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["value"] > 1e9, "north_star: >= 1e9 evals/s on one MI355X"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["launches_timed"] == 16
    # C3, 8 label keys; the sampled pick rides in the launch: + 20 B of draws per pod read, + 4 B of binding per pod written
    assert c["pick_launch"] in ("fused", "fused-tile") and c["kernels_per_step"] == 1
    assert r["algorithmic_bytes_per_launch"] == 100_000 * (48 + 20) + 5_000 * 48 + 100_000 * 79 * 8 + 100_000 * 4
    assert c["mask_rotation"] >= 5 and c["mask_rotation_bytes"] > 256 * 2**20, "the timed loop must not rewrite a mask the Infinity Cache still holds"
    assert d["ramp_steps"] >= 16 and d["untimed_steps_before_timed_region"] == d["warmup"] + d["ramp_steps"]
    # ... nor evaluate a pod batch whose operands the previous step left in L2: six different resident batches in turn
    ir = c["input_rotation"]
    assert ir["batches"] == 6 and ir["bytes_resident"] == 6 * 100_000 * 68 > 8 * 4 * 2**20
    for leg in ("in_place", "two_batches_in_flight"):
        assert c[leg] and "error" not in c[leg] and c[leg]["ms_per_step"] > 0, leg
    assert c["two_batches_in_flight"]["bindings_equal_sequential"] is True
    for wl in ("C4s", "C5s"):
        o = c["other_workloads"][wl]
        assert "error" not in o and o["value"] > 1e12 and 0.2 < o["mask_kernel_frac"] < 1.0 and o["pick_alone_us_per_step"] > 0, wl
    q = c["other_workloads"]["C3x4"]  # four queued batches passed as one call (the mask kernel's fill is paid once; round 6: the tile-test pick rides here too)
    assert "error" not in q and q["us_per_100k_pods"] > 0 and q["pick_in_mask_launch"] is True, q
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-9 and r["frac"] >= 0.38, "north_star: >= 40 % of the HBM roofline (0.40-0.42 measured with rotated outputs; 5 % slack for the box)"
    assert r["min_kernel_us"] <= r["median_kernel_us"] <= r["max_kernel_us"]
    # the HBM traffic of the mask kernel is measured by this very invocation: separate rocprofv3 --pmc passes, calibrated;
    # no wasted re-reads -- within 1.0 .. 1.25 x the algorithmic bytes -- and the committed figure of an earlier session beside it
    live = r["traffic_live"]
    if not live or "error" in live:  # (a fresh box pages the profiler in on first use: one more try before calling it broken)
        r2 = run_bench("--steps", "20", "--warmup", "2", "--ramp-ms", "2", "--kernel-samples", "4", "--no-cpu-baseline", "--no-others", "--repeats", "0", "--live-traffic", "on")["roofline"]
        assert r2["traffic_live"] and "error" not in r2["traffic_live"], (live, r2["traffic_live"])
        r = dict(r, traffic=r2["traffic"], traffic_live=r2["traffic_live"], traffic_source=r2["traffic_source"])
        live = r["traffic_live"]
    assert r["traffic_source"].startswith("this invocation") and r["traffic"] == live["hbm_bytes_per_launch"]
    assert 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.25, (r["traffic"], r["algorithmic_bytes_per_launch"])
    assert live["dispatches"][0] >= 10 and live["write_bytes"] >= 100_000 * 79 * 8
    assert r["traffic_committed"] is None or abs(r["traffic_committed"] - r["traffic"]) < 0.1 * r["traffic"]
    # SURVEY.md 8d "separately report end-to-end including H2D/D2H" (VERDICT r5 item 2): host arrays -> bindings / -> mask through ksched_eval, and
    # corev1 objects -> reconcile_batch of the C++ host mirror; beside the metric, never in `value`
    e = c["end_to_end"]
    assert e and "error" not in e and e["host_cores"] >= 1
    hb, hm, ob = e["host_arrays_to_bindings"], e["host_arrays_to_mask"], e["objects"]
    assert 0 < hb["ms_per_batch"] < hm["ms_per_batch"] and hb["calls"] >= 20 and abs(hb["evals_per_s"] - 5e8 / (hb["ms_per_batch"] * 1e-3)) < 1e-6 * hb["evals_per_s"]
    assert hb["evals_per_s"] < d["value"], "the PCIe-inclusive rate is never the metric"
    # the mask copied back: into result arrays kept from call to call (the figure), and into a fresh array per call (first touched by the copy) beside it
    assert hm["output"].startswith("result arrays kept") and hm["fresh_output_ms_per_batch"] > hm["ms_per_batch"] > 63.2 / 64.0, hm  # (63.2 MB cannot cross a 64 GB/s link in under a millisecond)
    assert "error" not in ob and ob["batches"] == 5 and 0 < ob["best_ms_per_batch"] <= ob["median_ms_per_batch"] and 20_000 < ob["pods_bound"] < 80_000
    assert ob["best_ms_per_batch"] < 40.0, "objects -> bindings -> snapshot for a C3-size batch: 47 ms in round 3, 15 - 21 ms on the 256-thread boxes of round 6"
    sp = ob["median_split"]
    assert sp and sp["draws_encode_device_ms"] > 0 and sp["warn_lines_ms"] < 1.0
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["unit"] == "evals/s" and b["cores"] >= 1 and b["value"] > 0 and "sample" in b
    # the run checks what it timed: the last timed step's bindings -- every pod -- and >= 4096 mask rows against the oracle
    pc = d["parity_check"]
    assert pc == c["parity_check"] and pc["mismatches"] == 0 and pc["bindings"] == 100_000 and pc["rows"] >= 4096 and pc["words"] == pc["rows"] * 79, pc
    assert pc["partial"] is False and pc["bindings_of"] == 100_000 and 0 <= pc["input_batch"] < 6  # (the batch the LAST timed step evaluated is the one checked)


def test_a_wrong_result_fails_the_bench(built):
    """The self-check is live: with a kernel ablation bit set (KSCHED_OPT_DEBUG: results invalid by definition -- bit 1 skips the rank
    searches) the line reports mismatches and the process exits non-zero."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--ramp-ms", "2", "--kernel-samples", "4",
                        "--no-cpu-baseline", "--no-others", "--repeats", "0", "--debug", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, "a bench whose kernel skips the rank searches must not pass its self-check"
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["parity_check"]["mismatches"] > 0 and "self-check FAILED" in r.stderr


def test_an_unverified_number_is_not_reported_as_a_good_one(built, tmp_path):
    """when the checker itself cannot run (here: no oracle library to load) the line says so AND the process exits 3, unless --allow-unchecked."""
    import shutil
    fake_root = tmp_path / "repo"
    fake_root.mkdir()
    for item in ("bench.py", "kube_scheduler_rs_reference_amd", "profiles"):
        src = os.path.join(ROOT, item)
        (shutil.copytree if os.path.isdir(src) else shutil.copy)(src, fake_root / item)
    (fake_root / "oracle").mkdir()  # an `oracle` package without its library: the checker raises on import
    (fake_root / "oracle" / "__init__.py").write_text("")
    args = [sys.executable, str(fake_root / "bench.py"), "--steps", "20", "--warmup", "2", "--ramp-ms", "2", "--kernel-samples", "4", "--no-cpu-baseline", "--no-others", "--repeats", "0"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and "could NOT RUN" in r.stderr, (r.returncode, r.stderr[-800:])
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["parity_check"]["mismatches"] is None and "error" in line["parity_check"]
    r = subprocess.run(args + ["--allow-unchecked"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0


def test_other_workloads_and_bindings_only(built):
    d = run_bench("--workload", "C2", "--steps", "50", "--warmup", "5", "--ramp-ms", "5", "--kernel-samples", "8", "--no-cpu-baseline", "--live-traffic", "off")
    assert d["config"]["nodes"] == 1000 and d["config"]["predicates"] == "FIT" and d["cpu_baseline"] is None
    assert d["roofline"]["traffic_live"] is None and (d["roofline"]["traffic"] is None or d["roofline"]["traffic_source"].startswith("profiles/pmc_traffic.json"))
    d = run_bench("--steps", "50", "--warmup", "5", "--ramp-ms", "5", "--kernel-samples", "8", "--no-cpu-baseline", "--no-mask")
    assert d["config"]["mask_written"] is False and d["config"]["kernel"] == "none"


def test_multi_gpu_path_in_a_one_rank_group(built):
    """The N > 1 code path -- RCCL process group, pod-row shards, the all-gather of the bindings behind the C ABI, ksched_pipe -- in a
    ONE-rank group (KSCHED_BENCH_FORCE_DIST=1; all a one-GPU box can run): the line carries the gather cadence and its two reference
    points, and the strong-scaling leg of configs[3]."""
    d = run_bench("--steps", "20", "--warmup", "5", "--ramp-ms", "5", "--kernel-samples", "8", "--no-cpu-baseline",
                  env={"KSCHED_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    c = d["config"]
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and c["pods_per_gpu"] == 100_000 and c["nodes"] == 5_000  # the same workload at every N
    # north_star's step: one all-gather per batch is the graded form; four batches per gather is the secondary figure
    assert c["steps_per_allgather"] == 1 and c["steps_in_flight"] >= 2 and c["allgather"].startswith("ksched_allgather_bindings")
    assert c["pipe_mode"].startswith("alternate")
    assert c["mask_rotation_bytes"] > 256 * 2**20 and c["steps_in_flight"] % 2 == 0, "the N > 1 loop's slots must exceed the Infinity Cache too"
    for leg in ("allgather_every_4", "no_allgather"):
        assert c[leg] and c[leg]["ms_per_step"] > 0, leg
    assert c["allgather_every_4"]["steps_per_allgather"] == 4
    assert d["parity_check"]["mismatches"] == 0 and d["parity_check"]["mismatches_all_ranks"] == 0 and d["parity_check"]["bindings"] == 100_000
    assert 0.3 < c["scaling_efficiency_vs_no_allgather"] < 1.3
    s = c["configs3_strong"]
    assert s and "error" not in s and s["pods_total"] == 1_000_000 and s["nodes"] == 10_000 and s["value"] > 1e12
    assert 0.05 < c["bound_fraction"] < 1.0


FAKE_RCCL = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
TEST_LIB = os.path.join(ROOT, "tests", "cpp", "hooks", "libksched_hip.so")  # the test build of the evaluator library: the only one with the hooks (tests/cpp/test_hooks.cpp)


def test_two_ranks_on_one_gpu_run_the_n_greater_one_code_end_to_end(built):
    """`python bench.py --gpus 2`, started plainly, on a ONE-GPU box (test hook KSCHED_BENCH_ONE_GPU=1 under KSCHED_TEST_HOOKS=1): the file launches itself under
    torch.distributed.run, both ranks sit on device 0, the communicator behind the C ABI is created with nranks = 2 over the RCCL stand-in (a blocking
    all-gather through shared memory; RCCL itself refuses one device twice), and everything the driver's N = 2, 4, 8 runs execute runs: row shards with
    lo > 0, ksched_allgather_bindings per batch, the table check across ranks, the every-4 and no-gather legs, configs[3] split two ways.  The value is
    not a scaling figure and the line says so."""
    assert os.path.exists(FAKE_RCCL), "make host"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(KSCHED_TEST_HOOKS="1", KSCHED_LIB=TEST_LIB, KSCHED_RCCL_LIB=FAKE_RCCL, KSCHED_BENCH_ONE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--ramp-ms", "5", "--kernel-samples", "8",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "re-launching as" in r.stderr and "--nproc-per-node=2" in r.stderr
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and c["pods_per_gpu"] == 100_000 and c["pods_total"] == 200_000 and "one_gpu_stand_in" in c
    assert c["allgather"].startswith("ksched_allgather_bindings") and c["steps_per_allgather"] == 1
    pc = d["parity_check"]
    assert pc["mismatches"] == 0 and pc["mismatches_all_ranks"] == 0 and pc["ranks_unchecked"] == 0 and pc["bindings"] == 100_000
    assert pc["gathered_table"] == {"rows": 200_000, "shards_hashed": 2, "shard_sums_differing_between_ranks": 0}
    assert abs(d["value"] - 200_000 * 5_000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    for leg in ("allgather_every_4", "no_allgather"):
        assert c[leg] and c[leg]["ms_per_step"] > 0, leg
    s = c["configs3_strong"]
    assert s and "error" not in s and s["pods_total"] == 1_000_000 and "500000 pods on this rank" in s["workload"]


def test_the_one_gpu_hook_is_refused_without_the_test_switch(built):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KSCHED_TEST_HOOKS")}
    env.update(KSCHED_BENCH_ONE_GPU="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "is a test hook" in (r.stderr + r.stdout)


def test_a_rank_whose_gathered_table_differs_fails_the_bench(built):
    """Every rank checks its OWN rows against the oracle; the other ranks' rows it only has from the all-gather.  With the stand-in told to hand rank 1 one
    wrong word in rank 0's part of the table, both ranks' own rows are still right -- the per-shard sums compared across ranks are what notices, and the
    run exits non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(KSCHED_TEST_HOOKS="1", KSCHED_LIB=TEST_LIB, KSCHED_RCCL_LIB=FAKE_RCCL, KSCHED_BENCH_ONE_GPU="1", FAKE_RCCL_CORRUPT_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--ramp-ms", "5", "--kernel-samples", "8",
                        "--no-cpu-baseline", "--no-strong-leg", "--repeats", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0, "a table that differs between the ranks must fail the run"
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    pc = d["parity_check"]
    assert pc["mismatches"] == 0 and pc["gathered_table"]["shard_sums_differing_between_ranks"] > 0 and pc["mismatches_all_ranks"] > 0
