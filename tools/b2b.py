#!/usr/bin/env python
"""Diagnostics: back-to-back launches of the mask kernel alone (no pick), timed as total/N with stream events,
next to the per-dispatch timing.  usage: python tools/b2b.py [--workload C3] [--debug BITS] [--packed]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth
from bench import WORKLOADS
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3"); ap.add_argument("--pods", type=int, default=None)
ap.add_argument("--debug", type=int, default=0); ap.add_argument("--packed", action="store_true")
ap.add_argument("--reps", type=int, default=50); ap.add_argument("--nodes", type=int, default=None)
a = ap.parse_args()
cfg, P, N, flag_names, pick, desc = WORKLOADS[a.workload]
P = a.pods or P
N = a.nodes or N
c = synth.make_config(cfg, P=P, N=N)
flags = sum(getattr(L, f) for f in flag_names)
dev = torch.device("cuda:0")
ev = Evaluator(0); ev.set_kernel("fused"); ev.set_nodes(**c.node_columns())
if a.debug: ev.set_option(L.OPT_DEBUG, a.debug)
t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).view(dt)).to(dev)
d_cpu, d_mem = t(c.req_cpu, np.int64), t(c.req_mem, np.int64)
d_sel = t(c.pod_sel, np.int32) if c.n_keys else None
d_tol = t(c.pod_tol, np.int64) if "TAINT" in flag_names else None
mask = ev.alloc_mask(P, pitched=not a.packed)
run = lambda: ev.eval_device(d_cpu, d_mem, d_sel, d_tol, None, flags, out_feasible=mask)
for i in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.reps): run()
e1.record(); torch.cuda.synchronize()
b2b = e0.elapsed_time(e1) * 1e3 / a.reps
ev.set_timing(True); ev.kernel_time_ms()
for i in range(a.reps): run()
torch.cuda.synchronize()
ms, n = ev.kernel_time_ms()
print(f"{a.workload} P={P} debug={a.debug} pitch={mask.stride(0)}: back-to-back {b2b:.1f} us/launch, per-dispatch {ms*1e3/n:.1f} us")
