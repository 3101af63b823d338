#!/usr/bin/env python
"""Several batches in flight, each on a share of the chip ("let the fill of launch i + 1 hide under the stores of launch i").

A block of the fused mask kernel owns a compute unit, so a launch that takes all 256 of them leaves the next batch's launch nowhere to fill while
it stores.  This sweep times, for one workload, the pipe's alternate mode over k streams (KSCHED_OPT_PIPE_MODE = k) with every launch kept to
`cus` compute units (KSCHED_OPT_GRID_CUS): k = 1 is the loop on one stream (bench.py's N = 1 form), strictly sequential; a third field asks for
at least that many output buffers in the rotation (session r5b: the step time depends on it).  Outputs rotate over
enough mask buffers to exceed the Infinity Cache; every configuration's last bindings are compared with the sequential loop's.

    python tools/inflight_sweep.py [workload=C3] [steps=2000] [short=20] [fused_pick=1]
prints one line per (k, cus): us per step over `steps` steps, and over regions of `short` steps (the driver's form: median / min of 15 regions).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (WORKLOADS, algorithmic_bytes, rotation_for)
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
short = int(sys.argv[3]) if len(sys.argv) > 3 else 20
fused_pick = int(sys.argv[4]) if len(sys.argv) > 4 else 1
grid = os.environ.get("SWEEP", "1:0,2:0,2:128,2:160,3:88,3:128,4:64,4:96,4:128,6:48,8:32")
cfg, P, N, flag_names, pick, desc = bench.WORKLOADS[name]
c = synth.make_config(cfg, P=P, N=N)
dev = torch.device("cuda:0")
ev = Evaluator(0)
ev.set_option(L.OPT_FUSED_PICK, fused_pick)
ev.set_nodes(**c.node_columns())
flags = sum(getattr(L, f) for f in flag_names) | (L.PICK_SAMPLED if pick == "sampled" else L.PICK_BESTFIT)
taint = "TAINT" in flag_names
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
d = (t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32) if c.n_keys else None, t(c.pod_tol, np.int64) if taint else None,
     t(c.samples, np.int32) if pick == "sampled" else None)
m0 = ev.alloc_mask(P)
mask_bytes = int(m0.stride(0)) * 8 * P
R = bench.rotation_for(mask_bytes, True)
n_keys = c.n_keys if "SEL" in flag_names else 0
alg = bench.algorithmic_bytes(P, N, n_keys, taint, pick_attempts=5 if pick == "sampled" else 0)
print(f"# {desc}\n# mask {mask_bytes / 2**20:.0f} MiB, rotation >= {R} buffers, algorithmic bytes per step {alg}, steps {steps}, short regions of {short}")

masks_pool = [m0]
ref = None


def region(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


for item in grid.split(","):
    k, cus, *rest = (int(x) for x in item.split(":"))
    extra = rest[0] if rest else 0  # "k:cus:slots": at least that many output buffers in the rotation
    ev.set_option(L.OPT_GRID_CUS, cus)
    if k == 1:
        depth = max(R, extra)
    else:
        depth = -(-max(R, k, extra) // k) * k  # a multiple of k: a slot keeps its stream
    while len(masks_pool) < depth:
        masks_pool.append(ev.alloc_mask(P))
    outs = [torch.full((P,), -2, dtype=torch.int32, device=dev) for _ in range(depth)]
    if k == 1:
        run = ev.bind_eval_device(*d, flags, out_feasible=masks_pool[:depth], out_bindings=outs)
        fn = lambda i: run(i % depth, i % depth)  # noqa: E731
        pipe = None
    else:
        pipe = ev.pipe(depth)
        ev.set_option(L.OPT_PIPE_MODE, k)
        sub = pipe.bind(*d, flags, masks_pool[:depth], outs)
        fn = lambda i: sub(i % depth)  # noqa: E731
    region(fn, 4 * depth)
    region(fn, 3000)  # clock ramp
    long_us = min(region(fn, steps) for _ in range(3))
    shorts = sorted(region(fn, short) for _ in range(15))
    torch.cuda.synchronize()
    got = outs[(short - 1) % depth].cpu()
    if ref is None:
        ref = got
    same = bool(torch.equal(got, ref)) and all(bool(torch.equal(o.cpu(), ref)) for o in outs)
    print(f"k={k} cus={cus or 256:3d} slots={depth:2d} pick={ev.last_pick:10s} long {long_us:6.2f} us/step (step_frac {alg / long_us / 1e3 / 8000:.3f})  "
          f"short[{short}] median {shorts[7]:6.2f} min {shorts[0]:6.2f} max {shorts[-1]:6.2f}  bindings_equal={same}", flush=True)
    if pipe is not None:
        pipe.close()
    ev.set_option(L.OPT_PIPE_MODE, 0)
ev.close()
