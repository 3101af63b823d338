#!/usr/bin/env python
"""Cost of the snapshot calls at the BASELINE node counts, on the GPU box.  Two numbers per call: `host` = how long the call
keeps the host (it does not wait for the device), `ready` = call + torch.cuda.synchronize(): until the device has finished the
build.  Also: the lazy best-fit rebuild (first PICK_BESTFIT evaluation after a change vs the second), and a steady loop of
[1-node update + bindings-only evaluation] against the same loop without updates."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth


def timed(f, reps=1):
    best_h, best_r = 1e9, 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        best_h, best_r = min(best_h, t1 - t0), min(best_r, t2 - t0)
    return best_h * 1e6, best_r * 1e6


SIZES = {"C3": 5_000, "C4": 10_000, "C5": 50_000}
WARMED = [False]
for cfg in (sys.argv[1:] or ["C3", "C4", "C5"]):
    N = SIZES[cfg]
    c = synth.make_config(cfg, P=20_000, N=N)
    ev = Evaluator(0)
    cols = c.node_columns()
    ev.set_nodes(**cols)  # first call: allocations
    sh, sr = timed(lambda: ev.set_nodes(**cols), reps=5)
    one = np.array([N // 2], np.uint32)
    k = [0]
    def upd1():
        k[0] += 1
        ev.update_nodes(one, c.avail_cpu[one] - k[0], c.avail_mem[one] - k[0])
    uh, ur = timed(upd1, reps=20)
    many = np.arange(0, N, 7, dtype=np.uint32)
    mh, mr = timed(lambda: ev.update_nodes(many, c.avail_cpu[many] - 1, c.avail_mem[many] - 1), reps=5)
    # lazy best-fit rebuild
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
    d_cpu, d_mem, d_sel = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32)
    d_tol = t(c.pod_tol, np.int64) if c.n_taints else None
    d_smp = t(c.samples, np.int32)
    bind = torch.empty((c.P,), dtype=torch.int32, device=dev)
    fl = L.FIT | L.SEL | (L.TAINT if c.n_taints else 0)
    bf = lambda: ev.eval_device(d_cpu, d_mem, d_sel, d_tol, None, fl | L.PICK_BESTFIT, out_binding=bind)
    upd1(); _, b_first = timed(bf); _, b_second = timed(bf, reps=3)
    # steady loop: [1-node update + sampled pick of 20k pods]
    sp = lambda: ev.eval_device(d_cpu, d_mem, d_sel, d_tol, d_smp, fl | L.PICK_SAMPLED, out_binding=bind)
    def loop(with_update, n=200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            if with_update: upd1()
            sp()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    # (the HIP runtime stalls ONCE per process for ~37 ms somewhere around its 600th update of this kind -- tools/stream_probe.py --;
    # run past that point before the clock starts, or one configuration of the process reports 210-230 us per iteration)
    loop(True, 900 if not WARMED[0] else 20); WARMED[0] = True
    l_u, l_0 = min(loop(True), loop(True)), loop(False)
    print(f"{cfg} N={N}: ksched_set_nodes host {sh:.0f} us, ready {sr:.0f} us | ksched_update_nodes(1 node) host {uh:.0f} us, ready {ur:.0f} us | "
          f"({len(many)} nodes, every tile) host {mh:.0f} us, ready {mr:.0f} us | best-fit eval of {c.P} pods: first after a change {b_first:.0f} us, "
          f"then {b_second:.0f} us | loop [update 1 node + sampled pick]: {l_u:.1f} us/iter vs {l_0:.1f} without updates")
    ev.close()
