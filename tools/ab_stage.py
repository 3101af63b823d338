#!/usr/bin/env python
"""In-process interleaved A/B of the mask kernel's fill variants (KSCHED_OPT_DEBUG bits of a -DKSCHED_STAGE_EXPERIMENTS=1 build).

  bash tools/build_variants.sh stage="-DKSCHED_STAGE_EXPERIMENTS=1"
  KSCHED_LIB=$PWD/build/variants/libksched_hip_stage.so python tools/ab_stage.py --workloads C3 C4s --rounds 4

One process, one snapshot per workload; every round visits every variant (interleaved, so clock and box drift hit all of them
alike): `--warm` untimed steps, then `--samples` steps with HIP events on every mask kernel dispatch.  Every variant's mask is
compared word for word with the shipped variant's before it is timed.  A phase trace (per-block timestamps) of every variant
follows the table."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth  # noqa: E402

NAMES = {0: "shipped (LDS-DMA)", 1 << 20: "vgpr", (1 << 20) | (1 << 23): "vgpr+prefetch", (1 << 20) | (1 << 21) | (1 << 23): "vgpr+prefetch+rot",
         1 << 21: "dma+rot", 1 << 22: "dma nt", (1 << 20) | (1 << 23) | (1 << 24): "aux vgpr+prefetch, rows dma"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="+", default=["C3", "C4s"])
    ap.add_argument("--variants", nargs="+", type=int, default=list(NAMES))
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--warm", type=int, default=150)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--no-trace", action="store_true")
    ap.add_argument("--allow-diff", action="store_true", help="time variants whose mask differs too (ablation bits)")
    a = ap.parse_args()
    print("library:", L.LIB_PATH)
    dev = torch.device("cuda:0")
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).view(dt)).to(dev)  # noqa: E731
    for wl in a.workloads:
        cfg, P, N, flag_names, pick, _ = WORKLOADS[wl]
        c = synth.make_config(cfg, P=P, N=N)
        flags = sum(getattr(L, f) for f in flag_names)
        ev = Evaluator(0)
        ev.set_kernel("fused")
        ev.set_nodes(**c.node_columns())
        d_cpu, d_mem = t(c.req_cpu, np.int64), t(c.req_mem, np.int64)
        d_sel = t(c.pod_sel, np.int32) if c.n_keys else None
        d_tol = t(c.pod_tol, np.int64) if "TAINT" in flag_names else None
        mask = ev.alloc_mask(P, pitched=True)
        ref = None

        def run(n):
            for _ in range(n):
                ev.eval_device(d_cpu, d_mem, d_sel, d_tol, None, flags, out_feasible=mask)

        # clock ramp
        run(3000)
        torch.cuda.synchronize()
        res = {v: [] for v in a.variants}
        for r in range(a.rounds):
            for v in a.variants:
                ev.set_option(L.OPT_DEBUG, v)
                if r == 0:
                    mask.zero_()
                    run(1)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = mask.clone()
                    elif not torch.equal(mask, ref):
                        print(f"{wl} variant {v}: MASK DIFFERS from the shipped variant ({int((mask != ref).sum())} words)")
                        if not a.allow_diff:
                            res.pop(v)
                            continue
                if v not in res:
                    continue
                run(a.warm)
                torch.cuda.synchronize()
                ev.set_timing(True, every=1)
                run(a.samples)
                torch.cuda.synchronize()
                ev.kernel_time_ms()
                run(a.samples)
                torch.cuda.synchronize()
                s = np.sort(ev.kernel_time_samples(a.samples * 2) * 1e3)
                ev.set_timing(False)
                res[v].append((float(s.mean()), float(np.median(s)), float(s.min())))
        base = np.mean([x[1] for x in res[a.variants[0]]]) if a.variants[0] in res else None
        for v, rows in res.items():
            med = np.mean([x[1] for x in rows])
            print(f"{wl} {str(NAMES.get(v, v)):>20s} debug {v:>9d}: median-of-rounds {med:6.2f} us  ({'%+.2f' % (med - base) if base else '?'})  rounds "
                  + " ".join(f"{m:.2f}/{md:.2f}/{mn:.2f}" for m, md, mn in rows))
        if not a.no_trace:
            names = ["entry", "staged_issue", "barrier", "phase1", "group0", "loop_end", "drained"]
            for v in res:
                ev.set_option(L.OPT_DEBUG, v)
                run(5)
                torch.cuda.synchronize()
                ev.set_option(L.OPT_TRACE, 1)
                run(1)
                torch.cuda.synchronize()
                tr = ev.trace_read()
                ev.set_option(L.OPT_TRACE, 0)
                tr = tr[tr[:, 0] > 0]
                rel = (tr[:, :7].astype(np.int64) - np.int64(tr[:, 0].min())) * 0.01
                d = rel[:, 1:7] - rel[:, 0:6]
                xcc = tr[:, 7].astype(np.int64)
                print(f"{wl} trace {NAMES.get(v, v)}: " + " ".join(f"{names[i + 1]}+{np.median(d[:, i]):.2f}" for i in range(6))
                      + f" | entry med {np.median(rel[:, 0]):.2f} max {rel[:, 0].max():.2f} | drained med {np.median(rel[:, 6]):.2f} max {rel[:, 6].max():.2f}")
                print("     entry by XCC   " + " ".join(f"{np.median(rel[xcc == x, 0]):.2f}" for x in range(8)))
                print("     drained by XCC " + " ".join(f"{np.median(rel[xcc == x, 6]):.2f}" for x in range(8)))
        ev.close() if hasattr(ev, "close") else None


if __name__ == "__main__":
    main()
