#!/usr/bin/env python
"""Probe: the steady loop [ksched_update_nodes(1 node) -> bindings-only evaluation] for the 1st, 2nd, 3rd ... context created in one
process.  tools/host_costs.py had shown one configuration of a process at 212-232 us per iteration where the others take 48 us.
What this probe found (profiles/r02_m_snapshot_stream.txt): it is neither the context nor its stream -- ONE loop of the process, around
its ~600th update, carries a single ~37 ms stall of the HIP runtime (a pool growing, by the look of it), and no later loop does
(16 x 200 further iterations at 31-32 us each).  --repeat prints consecutive loops on every context; --user-stream evaluates on a
non-default torch stream."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth

c = synth.make_config("C3", P=20_000, N=5_000)
dev = torch.device("cuda", 0)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
d_cpu, d_mem, d_sel, d_smp = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.samples, np.int32)
bind = torch.empty((c.P,), dtype=torch.int32, device=dev)
one = np.array([2500], np.uint32)
keep = []
user = torch.cuda.Stream() if "--user-stream" in sys.argv else None
for i in range(int(os.environ.get("PROBE_N", "9"))):
    ev = Evaluator(0)
    keep.append(ev)  # contexts stay alive: the i-th context's stream is the i-th stream this loop creates
    ev.set_nodes(**c.node_columns())
    k = [0]
    def it(update):
        if update:
            k[0] += 1
            ev.update_nodes(one, c.avail_cpu[one] - k[0], c.avail_mem[one] - k[0])
        ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, L.FIT | L.SEL | L.PICK_SAMPLED, out_binding=bind)
    def loop(update, n=200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            it(update)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    def split(n=200):  # host time of the two calls, separately
        tu = te = 0.0
        torch.cuda.synchronize()
        for _ in range(n):
            k[0] += 1
            t0 = time.perf_counter(); ev.update_nodes(one, c.avail_cpu[one] - k[0], c.avail_mem[one] - k[0]); t1 = time.perf_counter()
            ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, L.FIT | L.SEL | L.PICK_SAMPLED, out_binding=bind); t2 = time.perf_counter()
            tu += t1 - t0; te += t2 - t1
        torch.cuda.synchronize()
        return tu / n * 1e6, te / n * 1e6
    with torch.cuda.stream(user) if user is not None else torch.cuda.stream(torch.cuda.current_stream()):
        loop(True, 20)
        lu, l0 = loop(True), loop(False)
        if "--repeat" in sys.argv:  # is the slow loop a property of the context, or a one-off of the process?
            print("   consecutive loops of 200 [update + pick] on this context:", " ".join(f"{loop(True):.0f}" for _ in range(16)), "us/iter")
        hu, he = split()
        print(f"context #{i + 1}: loop [update 1 node + sampled pick] {lu:6.1f} us/iter, without updates {l0:5.1f} | host time per call: update_nodes {hu:6.1f} us, eval_device {he:5.1f} us")
