#!/usr/bin/env python
"""Host-side cost of the snapshot calls (index build, incremental update) at the BASELINE node counts.  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_scheduler_rs_reference_amd import Evaluator, synth
for cfg, N in (("C3", 5_000), ("C4", 10_000), ("C5", 50_000)):
    c = synth.make_config(cfg, P=1000, N=N)
    ev = Evaluator(0)
    t0 = time.perf_counter(); ev.set_nodes(**c.node_columns()); t1 = time.perf_counter()
    idx = np.array([N // 2], np.uint32)
    ts = []
    for i in range(5):
        t2 = time.perf_counter(); ev.update_nodes(idx, c.avail_cpu[idx] - i, c.avail_mem[idx] - i); ts.append(time.perf_counter() - t2)
    idx = np.arange(0, N, 7, dtype=np.uint32)
    t3 = time.perf_counter(); ev.update_nodes(idx, c.avail_cpu[idx] - 1, c.avail_mem[idx] - 1); t4 = time.perf_counter()
    print(f"{cfg} N={N}: ksched_set_nodes {1e3*(t1-t0):.1f} ms | ksched_update_nodes(1 node) {1e3*min(ts):.2f} ms | ({len(idx)} nodes, every tile) {1e3*(t4-t3):.1f} ms")
    ev.close()
