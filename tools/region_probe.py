#!/usr/bin/env python
"""Why does a region of 20 steps read 20 - 21 us per step where 2000 steps read 18.2 - 19.1 and the kernels 18.3?  (bench.py's --steps 20 form, r5s - r5t:
the region takes ~410 us whatever the host's launch loop and wait mode are.)

The C3 loop of bench.py's N = 1 form (outputs rotated over six masks; inputs rotated over six resident pod batches or not), regions of K steps between
synchronize pairs, under different histories:
    back-to-back        regions one after the other (what tools/fixed_cost.py does)
    idle T ms           the host sleeps T ms before every region (the GPU sits idle)
    behind a burst      3000 steps (60 ms of load), synchronize, then the region -- bench.py's situation
and, behind a burst, the per-dispatch kernel durations INSIDE the region (HIP events on every dispatch).
    python tools/region_probe.py [K=20] [regions=15]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
regions = int(sys.argv[2]) if len(sys.argv) > 2 else 15
dev = torch.device("cuda:0")
cfg, P, N, flag_names, pick, desc = bench.WORKLOADS["C3"]
B = 6
cs = [synth.make_config(cfg, P=P, N=N, pod_offset=b * P) for b in range(B)]
ev = Evaluator(0)
ev.set_nodes(**cs[0].node_columns())
flags = sum(getattr(L, f) for f in flag_names) | L.PICK_SAMPLED
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
batches = [(t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), None, t(c.samples, np.int32)) for c in cs]
masks = [ev.alloc_mask(P) for _ in range(6)]
out = torch.full((P,), -1, dtype=torch.int32, device=dev)
bounds = [ev.bind_eval_device(*b, flags, out_feasible=masks, out_bindings=[out]) for b in batches]


def make_step(nb):
    k = [0]

    def step():
        bounds[k[0] % nb](0, k[0] % 6)
        k[0] += 1
    return step


def region(step, k=K):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / k * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6


def report(label, rows):
    a = np.array(rows)
    print(f"{label:46s} us/step median {np.median(a[:, 0]):6.2f} min {a[:, 0].min():6.2f} max {a[:, 0].max():6.2f}   launch loop {np.median(a[:, 1]):5.0f} us, closing wait {np.median(a[:, 2]):5.0f} us", flush=True)


for nb in (6, 1):
    step = make_step(nb)
    for _ in range(3000):
        step()
    print(f"# inputs: {nb} batch(es) in rotation; K = {K}; {regions} regions per line")
    report("2000-step regions", [region(step, 2000) for _ in range(3)])
    report("back-to-back", [region(step) for _ in range(regions)])
    for idle_ms in (0.1, 1.0, 10.0, 100.0):
        rows = []
        for _ in range(regions):
            torch.cuda.synchronize()
            time.sleep(idle_ms * 1e-3)
            rows.append(region(step))
        report(f"idle {idle_ms:g} ms before every region", rows)
    rows = []
    for _ in range(regions):
        for _ in range(3000):
            step()
        rows.append(region(step))
    report("behind a burst of 3000 steps", rows)
    rows = []
    for _ in range(regions):
        for _ in range(3000):
            step()
        torch.cuda.synchronize()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 200e-6:  # the host busy for 200 us, the GPU idle
            pass
        rows.append(region(step))
    report("behind a burst + 200 us of GPU idle (host busy)", rows)
    # per-dispatch durations inside a region behind a burst
    ev.set_timing(True, every=1)
    acc = []
    for _ in range(regions):
        ev.set_timing(False)
        for _ in range(3000):
            step()
        torch.cuda.synchronize()
        ev.set_timing(True, every=1)
        ev.kernel_time_samples(4096)
        region(step)
        acc.append(ev.kernel_time_samples(4096)[:K] * 1e3)
    ev.set_timing(False)
    acc = np.array([a for a in acc if a.shape[0] == K])
    if acc.size:
        print("   kernel us by position in the region (median over regions): " + " ".join(f"{x:.1f}" for x in np.median(acc, axis=0)), flush=True)
ev.close()
