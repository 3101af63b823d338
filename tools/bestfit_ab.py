#!/usr/bin/env python
"""A/B of the best-fit pick at the C5 shard (125k pods x 50k nodes, fit + sel + taints), bindings-only requests (GPU box).
For each KSCHED_OPT_DEBUG value on the command line: every pod's binding against the oracle, then the time of a bindings-only step.
usage: python tools/bestfit_ab.py [--pods P] [--steps K] [--stages 0|1|2] [--no-oracle] <debug value> ...        (0 = the shipped form;
bit 20 = 0x100000: one call with per-wave time stamps, the library reports where the two stages' waves spent their time)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kube_scheduler_rs_reference_amd import Evaluator, synth, _lib as L

args = sys.argv[1:]
P, steps, use_oracle, stages = 125_000, 300, True, 0
vals = []
while args:
    a = args.pop(0)
    if a == "--pods":
        P = int(args.pop(0))
    elif a == "--steps":
        steps = int(args.pop(0))
    elif a == "--stages":
        stages = int(args.pop(0))
    elif a == "--no-oracle":
        use_oracle = False
    else:
        vals.append(int(a, 0))
vals = vals or [0]
c = synth.make_config("C5", P=P)
flags = L.FIT | L.SEL | L.TAINT | L.PICK_BESTFIT
dev = torch.device("cuda:0")
ev = Evaluator(0)
ev.set_nodes(**c.node_columns())
ev.set_option(L.OPT_BESTFIT_STAGES, stages)  # 0 = by batch size, 1 = wave per pod, 2 = lane per pod + wave per handed-over pod
want = None
if use_oracle:
    from oracle import capi
    t0 = time.time()
    want = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, c.node_taints, c.req_cpu, c.req_mem, c.pod_sel, c.pod_tol, None, flags)[2]
    print(f"oracle: {P} x {c.N} in {time.time() - t0:.1f} s; {int((want < 0).sum())} pods without a node", flush=True)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
d = (t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.pod_tol, np.int64), None)
out = torch.full((P,), -1, dtype=torch.int32, device=dev)
run = ev.bind_eval_device(*d, flags, out_feasible=None, out_bindings=[out])
for v in vals:
    ev.set_option(L.OPT_DEBUG, v)
    out.fill_(-7)
    run(0, 0)
    torch.cuda.synchronize()
    ok = "unchecked" if want is None else ("bit-exact" if np.array_equal(out.cpu().numpy(), want) else
                                           f"MISMATCH at {int((out.cpu().numpy() != want).sum())} pods")
    if v & 0x100000:  # bit 20: the library prints where the waves of the two stages spent their time (stderr), once per call
        print(f"debug={v:#010x}  traced call   {ok}", flush=True)
        continue
    for _ in range(10):
        run(0, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(0, 0)
    torch.cuda.synchronize()
    print(f"debug={v:#010x}  bindings-only step {(time.perf_counter() - t0) / steps * 1e6:7.1f} us   pick={ev.last_pick}   {ok}", flush=True)
