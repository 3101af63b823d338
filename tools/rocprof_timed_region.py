#!/usr/bin/env python
"""rocprofv3 --kernel-trace of a bench.py command: the dominant kernel's durations BY PHASE of the command.  `--stats` averages every
launch of the kernel name -- the clock ramp before the timed region, the graded steps, the in-place / two-stream / other-workload
legs after it (overlapped launches take twice as long each) --; the figure that corresponds to `roofline.avg_kernel_us` is the mean
over the launches of the timed region.  usage: python tools/rocprof_timed_region.py <kernel_trace.csv> <bench line .json (printed by that same run)> [kernel name substring]"""
import csv, json, sys
import numpy as np

trace, line = sys.argv[1], json.load(open(sys.argv[2]))
name = sys.argv[3] if len(sys.argv) > 3 else None
rows = list(csv.DictReader(open(trace)))
if name is None:  # the kernel with the largest total time
    tot = {}
    for r in rows:
        tot[r["Kernel_Name"]] = tot.get(r["Kernel_Name"], 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = max(tot, key=tot.get)
rows = [r for r in rows if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
u, k, rep = int(line["untimed_steps_before_timed_region"]), int(line["steps"]), len(line.get("repeat_ms_per_step") or line["config"].get("repeat_ms_per_step") or [])
seg = {"ramp_and_warmup": (0, u), "timed_region": (u, u + k), "repeats": (u + k, u + k + rep * k), "legs_after": (u + k + rep * k, len(dur))}
out = {"kernel": name.split("(")[0], "launches": int(len(dur)), "all_launches_mean_us": float(dur.mean()), "all_launches_median_us": float(np.median(dur)),
       "bench_line_avg_kernel_us_in_this_profiled_run": line["roofline"]["avg_kernel_us"], "segments": {}}
for s, (a, b) in seg.items():
    d = dur[a:min(b, len(dur))]
    if len(d):
        out["segments"][s] = {"launches": int(len(d)), "mean_us": float(d.mean()), "median_us": float(np.median(d)), "max_us": float(d.max())}
print(json.dumps(out, indent=1))
