#!/usr/bin/env python
"""Per-candidate counter values of tools/alloc_probe.py's final phase.

usage: alloc_pmc_summary.py <log with the PLAN line> <rocprofv3 output dir of the same run> [more (log, dir) pairs of other counter passes ...]

The probe's final phase launches the mask kernel `per_candidate_launches` times into each candidate buffer in turn and nothing else afterwards, so the
LAST len(candidates) * per_candidate_launches dispatches of the mask kernel in the csv are those, in order.  Prints, per pass, one row per candidate:
allocation path, the HIP-event mean of that pass, the mean of every counter (summed over the counter's dimensions) per launch; then the correlation of
each counter with the event time over the candidates, and its ratio slowest / fastest candidate.
"""
import collections
import csv
import glob
import json
import os
import sys

import numpy as np


def plan_of(log):
    for line in open(log, errors="replace"):
        if line.startswith("PLAN "):
            return json.loads(line[5:])
    raise SystemExit(f"{log}: no PLAN line")


def dispatches(dirname, prefix):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if prefix not in name:
                continue
            d = acc.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))
            d[r["Counter_Name"]] += float(r["Counter_Value"])
            # a counter reported per instance (several rows per dispatch): keep the largest instance too -> imbalance = max * rows / sum
            d[r["Counter_Name"] + "#max"] = max(d.get(r["Counter_Name"] + "#max", 0.0), float(r["Counter_Value"]))
            d[r["Counter_Name"] + "#rows"] += 1.0
    for d in acc.values():
        for k in [k for k in d if k.endswith("#rows")]:
            base = k[:-5]
            if d[k] <= 1.0:
                del d[base + "#max"]
            else:
                d[base + "#imbalance"] = d[base + "#max"] * d[k] / d[base] if d[base] else 0.0
                del d[base + "#max"]
            del d[k]
    return [acc[k] for k in sorted(acc)]


def main():
    args = sys.argv[1:]
    out = {}
    for log, d in zip(args[0::2], args[1::2]):
        plan = plan_of(log)
        n, cands = plan["per_candidate_launches"], plan["candidates"]
        disp = dispatches(d, plan["kernel_prefix"])
        need = n * len(cands)
        if len(disp) < need:
            print(f"{d}: {len(disp)} dispatches of the mask kernel, the plan needs {need}")
            continue
        disp = disp[-need:]
        counters = sorted({c for x in disp for c in x})
        rows = []
        for ci, c in enumerate(cands):
            chunk = disp[ci * n:(ci + 1) * n]
            rows.append({k: float(np.mean([x[k] for x in chunk])) for k in counters})
        t = np.array([c["event_mean_us"] for c in cands])
        print(f"== {os.path.basename(os.path.normpath(d))}: {plan['workload']}, {len(cands)} candidates x {n} launches; counters: {' '.join(counters)}")
        print("   " + f"{'path':10s} {'cand':>4s} {'event us':>9s} " + " ".join(f"{k[-26:]:>26s}" for k in counters))
        for c, r in zip(cands, rows):
            print("   " + f"{c['name']:10s} {c['candidate']:4d} {c['event_mean_us']:9.2f} " + " ".join(f"{r[k]:26.1f}" for k in counters))
        fast, slow = int(np.argmin(t)), int(np.argmax(t))
        for k in counters:
            v = np.array([r[k] for r in rows])
            corr = float(np.corrcoef(t, v)[0, 1]) if v.std() > 0 and t.std() > 0 else float("nan")
            ratio = v[slow] / v[fast] if v[fast] else float("nan")
            print(f"   {k:44s} corr with time {corr:+.3f}   slowest/fastest candidate {ratio:8.3f}   ({v[fast]:.4g} -> {v[slow]:.4g})")
            out.setdefault(plan["workload"], {})[k] = {"corr_with_time": corr, "slowest_over_fastest": float(ratio), "fastest": float(v[fast]), "slowest": float(v[slow]),
                                                       "time_fastest_us": float(t[fast]), "time_slowest_us": float(t[slow])}
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
