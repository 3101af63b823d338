#!/usr/bin/env python
"""Diagnostics: per-block phase timeline of the fused mask kernel (KSCHED_OPT_TRACE).
usage: python tools/trace_fused.py [--workload C3] [--pods N] [--debug BITS]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth
from bench import WORKLOADS

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3"); ap.add_argument("--pods", type=int, default=None)
ap.add_argument("--rotate", type=int, default=1, help="cycle over this many different pod batches AND mask buffers before the traced launch (the bench's form: operands and outputs cold)")
ap.add_argument("--profile", action="store_true", help="library built with -DKSCHED_PROFILE=1: trace words 1..4 are wave 0's cycle totals");
ap.add_argument("--pick", action="store_true", help="with the sampled pick riding in the launch (KSCHED_OPT_FUSED_PICK)"); ap.add_argument("--debug", type=int, default=0); ap.add_argument("--packed", action="store_true"); ap.add_argument("--nodes", type=int, default=None); ap.add_argument("--kill", type=int, default=0, help="make the last K nodes infeasible")
a = ap.parse_args()
cfg, P, N, flag_names, pick, desc = WORKLOADS[a.workload]
P = a.pods or P
N = a.nodes or N
B = max(1, a.rotate)
c = synth.make_config(cfg, P=P * B, N=N)
flags = sum(getattr(L, f) for f in flag_names)
if a.kill:
    c.avail_cpu[-a.kill:] = -1
    c.avail_mem[-a.kill:] = -1
dev = torch.device("cuda:0")
ev = Evaluator(0); ev.set_kernel("fused"); ev.set_nodes(**c.node_columns())
if a.debug: ev.set_option(L.OPT_DEBUG, a.debug)
t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).view(dt)).to(dev)
bt = []
for b in range(B):
    r = slice(b * P, (b + 1) * P)
    bt.append((t(c.req_cpu[r], np.int64), t(c.req_mem[r], np.int64), t(c.pod_sel[:, r], np.int32) if c.n_keys else None,
               t(c.pod_tol[r], np.int64) if "TAINT" in flag_names else None, t(c.samples[r], np.int32) if a.pick else None))
masks = [ev.alloc_mask(P, pitched=not a.packed) for _ in range(B)]
d_out = None
if a.pick:
    flags |= L.PICK_SAMPLED
    d_out = torch.empty((P,), dtype=torch.int32, device=dev)
def launch(i):
    d_cpu, d_mem, d_sel, d_tol, d_smp = bt[i % B]
    ev.eval_device(d_cpu, d_mem, d_sel, d_tol, d_smp, flags, out_feasible=masks[i % B], out_binding=d_out)
for i in range(5 * B):
    launch(i)
torch.cuda.synchronize()
ev.set_option(L.OPT_TRACE, 1)
for i in range(5 * B, 6 * B + 1):  # (every launch is traced; the buffer keeps the last one: a launch in the middle of the rotation)
    launch(i)
torch.cuda.synchronize()
tr = ev.trace_read()
live = tr[:, 0] > 0
tr = tr[live]
pick_done = (tr[:, 7] >> np.uint64(8)).astype(np.float64) * 0.01  # wave 0: entry -> picks done (us); 0 without a riding pick
tr[:, 7] &= np.uint64(0xFF)
if a.pick:
    print(f"pick: {ev.last_pick}; wave 0 entry -> its picks done: median {np.median(pick_done):.2f} p90 {np.percentile(pick_done, 90):.2f} max {pick_done.max():.2f} us (includes its wait at the block's barrier)")
if a.profile:
    p2, wt, p1, rounds = (tr[:, i].astype(np.float64) for i in (1, 2, 3, 4))
    life = (tr[:, 6].astype(np.float64) - tr[:, 0].astype(np.float64)) * 0.01  # us, block entry -> drained
    print(f"{a.workload} P={P} N={N} blocks={len(tr)}: wave 0 of every block, core cycles (s_memtime) -- medians")
    r = np.maximum(rounds, 1)
    print(f"  rounds per wave          {np.median(rounds):8.1f}")
    print(f"  phase 2 / round          {np.median(p2 / r):8.0f} cycles   total {np.median(p2):9.0f}")
    print(f"  operand wait / round     {np.median(wt / r):8.0f} cycles   total {np.median(wt):9.0f}")
    print(f"  phase 1 / round          {np.median(p1 / r):8.0f} cycles   total {np.median(p1):9.0f}")
    print(f"  block lifetime           {np.median(life):8.2f} us")
    sys.exit(0)
t0 = tr[:, 0].min()
rel = (tr[:, :7].astype(np.int64) - np.int64(t0)) * 0.01  # us (100 MHz)
names = ["entry", "staged_issue", "barrier", "phase1", "group0", "loop_end", "drained"]
print(f"{a.workload} P={P} N={N} blocks traced={len(tr)}  (times in us since the first block's entry)")
for i, n in enumerate(names):
    v = rel[:, i][tr[:, i] > 0]
    if len(v): print(f"  {n:13s} min {v.min():7.2f}  median {np.median(v):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f}")
d = rel[:, 1:7] - rel[:, 0:6]
for i in range(6):
    v = d[:, i]
    print(f"  d[{names[i]}->{names[i+1]}] median {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f} max {v.max():6.2f}")
print("  blocks per XCC:", np.bincount(tr[:, 7].astype(np.int64), minlength=8))
xcc = tr[:, 7].astype(np.int64)
print("  drained per XCC (median/max):", " ".join(f"{np.median(rel[xcc==x,6]):.1f}/{rel[xcc==x,6].max():.1f}" for x in range(8)))
h, e = np.histogram(rel[:, 6], bins=12)
print("  drained histogram:", " ".join(f"{e[i]:.1f}:{h[i]}" for i in range(len(h))))
ids = np.nonzero(live)[0]
tiles = ev.W // 16 + (1 if ev.W % 16 else 0)
tile_of = (ids >> 3) % tiles
for i, n in enumerate(names):
    print(f"  {n:13s} median by tile:", " ".join(f"{np.median(rel[tile_of==t, i]):.1f}" for t in range(tiles)))
late = np.argsort(-rel[:, 6])[:12]
print("  latest blocks (block id, tile, chunk, xcc, drained):", [(int(ids[i]), int((ids[i] >> 3) % tiles), int(((ids[i] >> 3) // tiles) * 8 + (ids[i] & 7)), int(xcc[i]), round(float(rel[i, 6]), 1)) for i in late])
