#!/usr/bin/env python
"""Can the best-fit pick run BESIDE the mask kernel of the same batch?  (the C5 shard's step is its 160 us mask kernel
plus 57 us of best-fit pick behind it.)

The two do not depend on each other (neither pick stage reads the mask), so they can sit on two streams -- but a 1024-thread block of the
mask kernel takes a CU's whole register file when it uses more than 104 VGPRs per lane (16 waves x 120 allocated = 480 of 512 per SIMD), and
then no wave of another kernel fits beside it.  This probe times, for the C5 shard,
    sequential : mask-only evaluation, then bindings-only best-fit evaluation, on ONE stream
    two streams: the same two calls on two streams
with whatever library KSCHED_LIB names (tools/build_variants.sh: e.g. -DKSCHED_FUSED_WPE=5 caps the mask kernel at 96 VGPRs), so that a
build whose mask kernel leaves registers free shows whether the picks then hide behind it.
usage: [KSCHED_LIB=...] python tools/coresident_probe.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
P, N = 125_000, 50_000
c = synth.make_config("C5", P=P, N=N)
dev = torch.device("cuda:0")
ev = Evaluator(0)
ev.set_nodes(**c.node_columns())
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
d_cpu, d_mem, d_sel, d_tol = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.pod_tol, np.int64)
mask = ev.alloc_mask(P, pitched=True)
bind = torch.empty((P,), dtype=torch.int32, device=dev)
preds = L.FIT | L.SEL | L.TAINT
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def mask_only(stream):
    ev.eval_device(d_cpu, d_mem, d_sel, d_tol, None, preds, out_feasible=mask, stream=stream)


def pick_only(stream):
    ev.eval_device(d_cpu, d_mem, d_sel, d_tol, None, preds | L.PICK_BESTFIT, out_binding=bind, stream=stream)


def timed(fn, k):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


pick_only(sa)  # (builds the best-fit structures)
torch.cuda.synchronize()
ref = bind.clone()
m = timed(lambda: mask_only(sa), steps)
p = timed(lambda: pick_only(sa), steps)
seq = timed(lambda: (mask_only(sa), pick_only(sa)), steps)
two = timed(lambda: (pick_only(sb), mask_only(sa)), steps)   # pick first: its first stage is short-lived waves that must get onto the chip
two2 = timed(lambda: (mask_only(sa), pick_only(sb)), steps)
ok = bool(torch.equal(bind, ref))
print(f"lib {os.environ.get('KSCHED_LIB', 'default')}: mask only {m:.1f} us | pick only {p:.1f} us | one stream {seq:.1f} us | two streams (pick enqueued first) {two:.1f} us | "
      f"two streams (mask first) {two2:.1f} us | bindings unchanged: {ok}")

# Per-CALL form (what a fork / join inside ksched_eval_device would be): every step is synchronised, so nothing of step i + 1 overlaps step i;
# the mask kernel kept to fewer compute units (KSCHED_OPT_GRID_CUS) leaves the pick's one-wave blocks room beside it.
if os.environ.get("PROBE_PER_CALL", "1") == "1":
    def timed_sync(fn, k):
        for _ in range(3):
            fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e6
    base = timed_sync(lambda: (mask_only(sa), pick_only(sa)), steps)
    print(f"per call (synchronised every step): one stream {base:.1f} us")
    for cus in (0, 236, 226, 216, 196):
        ev.set_option(L.OPT_GRID_CUS, cus)
        m1 = timed_sync(lambda: mask_only(sa), steps)
        t2 = timed_sync(lambda: (mask_only(sa), pick_only(sb)), steps)
        print(f"  mask kernel on {cus or 256} CUs: mask alone {m1:.1f} us | mask + pick on two streams {t2:.1f} us ({t2 - base:+.1f} vs one stream) | bindings unchanged: {bool(torch.equal(bind, ref))}")
    ev.set_option(L.OPT_GRID_CUS, 0)
