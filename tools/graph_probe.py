#!/usr/bin/env python
"""Probe: K sequential steps (pick + mask kernel through ksched_eval_device_pitched) captured into ONE hipGraph (torch.cuda.graph)
and replayed, against the same steps launched one library call at a time.  Launch-bound configs (C2) are where it matters."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth
dev = torch.device("cuda", 0)
for cfg, P, N, flags in (("C2", 10_000, 1_000, L.FIT | L.PICK_SAMPLED), ("C3", 100_000, 5_000, L.FIT | L.SEL | L.PICK_SAMPLED)):
    c = synth.make_config(cfg, P=P, N=N)
    ev = Evaluator(0)
    ev.set_nodes(**c.node_columns())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
    d_cpu, d_mem = t(c.req_cpu, np.int64), t(c.req_mem, np.int64)
    d_sel = t(c.pod_sel, np.int32) if c.n_keys else None
    d_smp = t(c.samples, np.int32)
    mask, bind = ev.alloc_mask(P), torch.empty((P,), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(device=dev)
    run = ev.bind_eval_device(d_cpu, d_mem, d_sel, None, d_smp, flags, out_feasible=mask, out_bindings=[bind], stream=s)
    K = 20
    with torch.cuda.stream(s):
        for _ in range(50): run(0)
    torch.cuda.synchronize()
    ref = bind.clone()
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(2000): run(0)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 2000 * 1e6
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            for _ in range(K): run(0)
        torch.cuda.synchronize()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / (100 * K) * 1e6
        ok = bool(torch.equal(bind, ref))
        print(f"{cfg}: one call per step {eager:.2f} us/step | hipGraph of {K} steps {graph:.2f} us/step | bindings equal: {ok}")
    except Exception as e:
        print(f"{cfg}: one call per step {eager:.2f} us/step | graph capture failed: {e}")
    ev.close()
