#!/usr/bin/env python
"""Does the mask kernel's rate depend on WHERE its output buffer lies?  (r5: the same kernel on the same box reads 140 us in one process and 174 us
in the next, flat in time within each -- tools/clock_ramp.py -- so it is not a clock ramp.)

One process, one evaluator, one workload: the mask buffer(s) are re-allocated several times -- behind spacers of different sizes, so that the virtual
and physical placement moves -- and carved out of their allocation at different byte offsets; every placement is timed with HIP events on the mask kernel
(mean of `n` dispatches after a warm-up).  Prints the device address next to the time.
    python tools/placement_probe.py [workload=C5s] [n=48] [offsets|sizes|sticky|select]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5s"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
rig = bench.SingleRig(torch, L, synth, Evaluator, dev, name)
ev, P = rig.ev, rig.P
m0 = ev.alloc_mask(P)
W, pitch = int(m0.shape[1]), int(m0.stride(0))
R = bench.rotation_for(pitch * 8 * P, True)
del m0
torch.cuda.empty_cache()
print(f"# {rig.desc}: mask rows of {W} words at a pitch of {pitch} words ({pitch * 8} B), {P} rows = {pitch * 8 * P / 2**20:.0f} MiB per buffer, {R} buffer(s) in rotation")


def timed(masks):
    run = ev.bind_eval_device(*rig.d, rig.flags, out_feasible=masks, out_bindings=[rig.out])
    k = 0
    for _ in range(max(12, 2 * len(masks))):
        run(0, k % len(masks)); k += 1
    torch.cuda.synchronize()
    ev.set_timing(True, every=1)
    for _ in range(n):
        run(0, k % len(masks)); k += 1
    torch.cuda.synchronize()
    ev.kernel_time_ms()
    for _ in range(n):
        run(0, k % len(masks)); k += 1
    torch.cuda.synchronize()
    us = ev.kernel_time_samples(2 * n) * 1e3
    ev.set_timing(False)
    return float(np.mean(us)), float(np.median(us)), float(np.min(us))


spacers = []
rows = P * pitch
mode = sys.argv[3] if len(sys.argv) > 3 else "offsets"
if mode == "select":
    # can the fast allocations be PICKED?  K candidate buffers alive at once, every one timed inside a rotation over all of them (so that no
    # buffer is rewritten while the Infinity Cache still holds it), twice; then the standard loop over the R fastest and over the R slowest
    K = max(3 * R, 6)
    cands = [ev.alloc_mask(P) for _ in range(K)]

    def per_buffer(reps=10):
        run = ev.bind_eval_device(*rig.d, rig.flags, out_feasible=cands, out_bindings=[rig.out])
        for i in range(2 * K):
            run(0, i % K)
        torch.cuda.synchronize()
        ev.set_timing(True, every=1)
        ev.kernel_time_samples(4096)
        for _ in range(reps):
            for i in range(K):
                run(0, i)
        torch.cuda.synchronize()
        us = ev.kernel_time_samples(4096) * 1e3
        ev.set_timing(False)
        return np.median(us.reshape(reps, K), axis=0)
    a, b = per_buffer(), per_buffer()
    print("per-buffer median us, pass 1: " + " ".join(f"{x:.1f}" for x in a))
    print("per-buffer median us, pass 2: " + " ".join(f"{x:.1f}" for x in b))
    print(f"correlation between the passes {np.corrcoef(a, b)[0, 1]:.2f}; spread of pass 1: min {a.min():.2f} median {np.median(a):.2f} max {a.max():.2f}")
    order = np.argsort(a + b)
    for label, idx in (("fastest", order[:R]), ("slowest", order[-R:]), ("fastest again", order[:R]), ("first allocated", np.arange(R))):
        mean, med, lo = timed([cands[i] for i in idx])
        print(f"loop over the {R} {label:16s} buffers {list(map(int, idx))}: mask kernel mean {mean:.2f} us, median {med:.2f}, min {lo:.2f}", flush=True)
    plan = []
elif mode == "sticky":
    # is the mode a property of the ALLOCATION?  One mask allocation timed repeatedly, with other allocations coming and going in between and the
    # operand columns re-allocated (cloned) half-way; then the next mask allocation
    import itertools
    for a in range(6):
        slabs = [torch.empty(rows, dtype=torch.int64, device=dev) for _ in range(R)]
        masks = [s_[:rows].view(P, pitch)[:, :W] for s_ in slabs]
        out = []
        for rep in range(6):
            if rep == 3:
                rig.d = tuple(None if t is None else t.clone() for t in rig.d)
                out.append("| operands re-allocated |")
            junk = torch.empty((17 + 40 * rep) << 20, dtype=torch.uint8, device=dev)
            out.append("%.1f" % timed(masks)[0])
            del junk
        print(f"mask allocation {a} at {masks[0].data_ptr():#x}: mask kernel mean us " + " ".join(out), flush=True)
        del masks, slabs
        torch.cuda.empty_cache()
        spacers.append(torch.empty((5 + 29 * a) << 20, dtype=torch.uint8, device=dev))
    plan = []
elif mode == "offsets":
    plan = [("exact", sp, off) for sp, off in [(0, 0), (0, 0), (3, 0), (0, 4096), (65, 0), (0, 65536), (513, 0), (0, 1 << 20), (1, 0), (0, 256), (0, 128), (0, 0)]]
else:  # "sizes": the slab the buffer is carved from rounded up to a power of two / a multiple of 256 MiB, against the exact size; no spacers
    plan = [(kind, 0, 0) for _ in range(5) for kind in ("exact", "pow2", "exact", "m256", "pow2x2")]


def slab_words(kind, need):
    b = need * 8
    if kind == "pow2":
        b = 1 << (b - 1).bit_length()
    elif kind == "pow2x2":
        b = 2 << (b - 1).bit_length()
    elif kind == "m256":
        b = -(-b // (256 << 20)) * (256 << 20)
    return b // 8


for trial, (kind, spacer_mb, off_bytes) in enumerate(plan):
    if spacer_mb:
        spacers.append(torch.empty(spacer_mb << 20, dtype=torch.uint8, device=dev))
    off = off_bytes // 8
    slabs = [torch.empty(slab_words(kind, rows + off), dtype=torch.int64, device=dev) for _ in range(R)]
    masks = [s[off:off + rows].view(P, pitch)[:, :W] for s in slabs]
    mean, med, lo = timed(masks)
    print(f"trial {trial:2d}: slab {kind:6s} {slabs[0].numel() * 8 / 2**20:6.0f} MiB, spacer {spacer_mb:4d} MiB, offset {off_bytes:8d} B, first buffer at {masks[0].data_ptr():#x}: "
          f"mask kernel mean {mean:7.2f} us, median {med:7.2f}, min {lo:7.2f}", flush=True)
    del masks, slabs
    torch.cuda.empty_cache()
rig.close()
