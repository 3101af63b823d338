#!/usr/bin/env python
"""How long does sustained load take to bring a store-bound kernel to its steady rate?  (r5: the C5 shard's mask kernel reads 140 us in one
process and 174 us in the next on the same box; the C3 kernel 17.8 against 20.4.)

For each workload: build the rig, leave the GPU idle for `idle` seconds (what generating a cluster on the host does), then step continuously for
`span` seconds in bursts of `burst` steps (one synchronize per burst) and print us per step against the time since the first launch; the DPM
tables of the memory / fabric / SoC / shader clocks (sysfs pp_dpm_*) are read at a few points of the run.
    python tools/clock_ramp.py [workloads=C3,C5s] [span=1.6] [idle=2.0] [burst=32]
"""
import glob
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth  # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "C3,C5s").split(",")
span = float(sys.argv[2]) if len(sys.argv) > 2 else 1.6
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
burst = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0")


def dpm():
    out = {}
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_*")):
        try:
            cur = [ln.strip() for ln in open(f) if "*" in ln]
            out[os.path.basename(f)[7:]] = cur[0] if cur else "?"
        except OSError:
            pass
    return out


for name in names:
    rig = bench.SingleRig(torch, L, synth, Evaluator, dev, name)
    step, R, masks = rig.loop(True)
    for _ in range(2 * R + 4):
        step()
    torch.cuda.synchronize()
    for rep in range(2):
        time.sleep(idle)
        print(f"# {name}: {rig.desc}; {R} mask buffers; idle {idle} s before; bursts of {burst} steps; dpm at idle {dpm()}", flush=True)
        series, marks = [], {}
        t0 = time.perf_counter()
        next_mark = 0.1
        while True:
            tb = time.perf_counter()
            for _ in range(burst):
                step()
            torch.cuda.synchronize()
            now = time.perf_counter()
            series.append((now - t0, (now - tb) / burst * 1e6))
            if now - t0 >= next_mark:
                marks[round(now - t0, 2)] = dpm()
                next_mark *= 2.2
            if now - t0 >= span:
                break
        # ten bursts per printed point: (ms since start, us per step)
        k = max(1, len(series) // 48)
        pts = [(series[i][0] * 1e3, sum(s[1] for s in series[i:i + k]) / len(series[i:i + k])) for i in range(0, len(series), k)]
        print("   " + "  ".join(f"{t:5.0f}ms:{u:6.2f}" for t, u in pts), flush=True)
        first = sum(s[1] for s in series[:3]) / 3
        last = sum(s[1] for s in series[-10:]) / 10
        settle = next((t for t, u in series if u <= 1.02 * last), None)
        print(f"   first bursts {first:.2f} us/step, last {last:.2f}; within 2 % of the last from {settle * 1e3:.0f} ms on; min burst {min(s[1] for s in series):.2f}", flush=True)
        for t, d in marks.items():
            print(f"   dpm at {t:5.2f} s: {d}", flush=True)
    del masks, step
    rig.close()
