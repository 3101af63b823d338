#!/usr/bin/env python
"""Copy the summaries of one tools/gpu_round.sh session from gpurun_out/<tag>/ (scratch) into profiles/ (tracked).
usage: python tools/collect_profiles.py <tag> [<round prefix, default r01>]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
pre = f"{rnd}_{tag}_"


def cp(name, out):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, pre + out))
        print("  ", pre + out)


cp("bench_default.json", "bench_default_C3.json")
cp("prof_default_kernel_stats.csv", "rocprofv3_kernel_stats_bench_default.csv")
cp("prof_default_bench.json", "bench_line_under_rocprofv3.json")
cp("pmc_traffic.json", "pmc_traffic.json")
cp("trace_C3.txt", "fused_phase_trace_C3.txt")
cp("trace_C3_nopick.txt", "fused_phase_trace_C3_without_pick.txt")
cp("bench_driver_form.json", "bench_driver_form_steps20_C3.json")
cp("prof_driver_form_kernel_stats.csv", "rocprofv3_kernel_stats_bench_driver_form.csv")
cp("prof_driver_form_bench.json", "bench_line_under_rocprofv3_driver_form.json")
cp("prof_c5s_kernel_stats.csv", "rocprofv3_kernel_stats_bench_C5s.csv")
cp("prof_default_by_phase.json", "rocprofv3_by_phase_default.json")
cp("prof_driver_form_by_phase.json", "rocprofv3_by_phase_driver_form.json")
cp("pytest_gpu.log", "pytest_gpu.log")
cp("smoke.log", "smoke.log")
cp("host_costs.txt", "host_costs_snapshot_calls.txt")
cp("fuzz.txt", "fuzz.txt")
cp("host_loop_C3.txt", "host_loop_C3_size_batch.txt")
cp("host_loop_small.txt", "host_loop_5k_and_20k_pods.txt")
cp("spread.json", "C3_spread_over_fresh_processes.json")
cp("box.txt", "box.txt")
cp("stream_probe.txt", "stream_probe.txt")
if os.path.exists(os.path.join(src, "pmc_traffic.json")):
    shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(dst, "pmc_traffic.json"))  # what bench.py reports as roofline.traffic
lines = {}
for f in sorted(glob.glob(os.path.join(src, "bench_C*.json"))):
    try:
        lines[os.path.basename(f)[6:-5]] = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        lines[os.path.basename(f)] = f"unreadable: {e}"
if lines:
    json.dump(lines, open(os.path.join(dst, pre + "bench_lines_" + "_".join(lines) + ".json"), "w"), indent=1)
    print("  ", pre + "bench_lines_*.json")
# SQ counters of the default workload: mean per dispatch and kernel
sq = glob.glob(os.path.join(src, "C3_sq", "**", "*counter_collection.csv"), recursive=True)
if sq:
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in sq:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]][r.get("Dispatch_Id", "0")] += float(r["Counter_Value"])
    out = {"command": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY "
                      "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline",
           "unit": "mean per dispatch, summed over the chip",
           "kernels": {k: {c: sum(v.values()) / len(v) for c, v in d.items()} for k, d in acc.items() if "ksched" in k}}
    json.dump(out, open(os.path.join(dst, pre + "sq_counters_C3.json"), "w"), indent=1)
    print("  ", pre + "sq_counters_C3.json")
