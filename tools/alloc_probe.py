#!/usr/bin/env python
"""Which allocation path gives the mask kernel its fast rate, and which hardware counter separates the two rates?
NEEDS THE TEST BUILD of the library for every path but 1 (plain) and 11 (probe): KSCHED_LIB=tests/cpp/hooks/libksched_hip.so (the measurement paths
live in tests/cpp/test_hooks.cpp; the shipped library answers KSCHED_E_UNSUPPORTED).
(the finding it follows up: profiles/r05_bimodal_by_allocation.md.)

    python tools/alloc_probe.py <workload=C5s> survey [k=4] [hows=1,2,4,5,8,9,11] [n=24]
        k mask buffers (sets of R buffers where the workload rotates) per allocation path of ksched_mask_alloc, each timed with HIP events on the
        mask kernel; a fragmenting pattern of spacer allocations runs between the paths so that the plain path sees used memory.
    python tools/alloc_probe.py <workload> pmc [k=6] [hows=1,2] [n=16]
        the same candidates; then a FINAL PHASE of n launches into each candidate in turn, and one `PLAN {json}` line that says which of the
        process's last launches wrote which candidate and what the events read -- run it under `rocprofv3 --kernel-trace --pmc <counters>` and give
        the csv to tools/alloc_pmc_summary.py: per-candidate counter values next to per-candidate times.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5s"
mode = sys.argv[2] if len(sys.argv) > 2 else "survey"
kw = dict(a.split("=", 1) for a in sys.argv[3:] if "=" in a)
K = int(kw.get("k", 4))
N = int(kw.get("n", 24))
hows = [int(h) for h in kw.get("hows", "1,2,4,5,8,9,11").split(",")]
frag = int(kw.get("frag", 1))
dev = torch.device("cuda:0")
rig = bench.SingleRig(torch, L, synth, Evaluator, dev, name, debug=int(kw.get("debug", "0"), 0))
ev, P = rig.ev, rig.P
pitch = int(ev._lib.ksched_mask_pitch(ev.n))
R = bench.rotation_for(pitch * 8 * P, True)
print(f"# {rig.desc}: {P} rows at a pitch of {pitch} words = {pitch * 8 * P / 2**20:.0f} MiB per buffer, {R} buffer(s) per candidate set", flush=True)


def bound(masks):
    return ev.bind_eval_device(*rig.d, rig.flags, out_feasible=masks, out_bindings=[rig.out])


def timed(masks, n=N):
    run = bound(masks)
    k = 0
    for _ in range(max(8, 2 * len(masks))):
        run(0, k % len(masks)); k += 1
    torch.cuda.synchronize()
    ev.set_timing(True, every=1)
    ev.kernel_time_samples(8192)
    for _ in range(n):
        run(0, k % len(masks)); k += 1
    torch.cuda.synchronize()
    us = ev.kernel_time_samples(8192) * 1e3
    ev.set_timing(False)
    return float(np.mean(us)), float(np.median(us)), float(np.min(us))


def fragment(seed):
    """Leave holes: allocate many odd-sized blocks through the plain path, free every other one.  Returned blocks stay alive."""
    rng = np.random.default_rng(seed)
    blocks = [torch.empty(int(rng.integers(3, 90)) << 20, dtype=torch.uint8, device=dev) for _ in range(48)]
    keep = blocks[::2]
    del blocks
    torch.cuda.empty_cache()
    return keep


if mode == "pitch":
    # Is the ROW PITCH a lever?  The stalls that separate the rates are HBM-side (TCC_EA0_WRREQ_DRAM_CREDIT_STALL, profiles/r06_mask_alloc.md): the
    # same 128-byte segments at another row stride meet the channel / bank hash differently.  Per pitch: k plain torch buffers, `passes` passes.
    W = ev.W
    pitch_how = int(kw.get("how", 0))  # 0: torch's allocator; else a ksched_mask_alloc path (e.g. 5 = one physically contiguous range)
    pitches = [int(x) for x in kw.get("pitches", "").split(",") if x] or sorted({pitch, pitch + 16, pitch + 32, pitch + 48, pitch + 64, pitch + 112, pitch + 128, -(-pitch // 128) * 128 + 16})
    table = {}
    for pw in pitches:
        def one(pw=pw):
            if not pitch_how:
                return torch.empty((P, pw), dtype=torch.int64, device=dev)[:, :W]
            rows_std = -(-P * pw // pitch)  # rows of the library's own pitch that cover P rows of this one
            return ev.alloc_mask(rows_std, how=pitch_how)._base.view(-1)[:P * pw].view(P, pw)[:, :W]
        sets = [[one() for _ in range(bench.rotation_for(pw * 8 * P, True))] for _ in range(K)]
        rows = np.array([[timed(m)[0] for m in sets] for _ in range(int(kw.get("passes", 2)))])
        table[pw] = rows
        print(f"pitch {pw:5d} words ({pw * 8:6d} B, {pw * 8 * P / 2**20:5.0f} MiB): per-candidate means us " + " | ".join(" ".join(f"{x:7.2f}" for x in r) for r in rows) +
              f"   => mean {rows.mean():7.2f}, fastest {rows.min():7.2f}, slowest {rows.max():7.2f}", flush=True)
        del sets
        torch.cuda.empty_cache()
    print("PITCH " + json.dumps({"workload": name, "W": W, "rows": {str(k_): v.tolist() for k_, v in table.items()}}))
    rig.close()
    sys.exit(0)

if mode == "ballast":
    # Does the rate follow WHERE IN VRAM the buffer lies?  A ballast of g GiB is allocated first (the driver hands out VRAM from one end), then k
    # candidates behind it, timed; then everything is freed and the next ballast size follows.  (288 GB = 8 stacks x 36 GB is not a power of two:
    # if the address map treats one end of the range differently, this finds the border.)
    how = int(kw.get("how", 1))
    for g in [int(x) for x in kw.get("gib", "0,16,32,64,96,128,160,192,224,240,256").split(",")]:
        try:
            ballast = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(g)]
        except Exception as e:  # noqa: BLE001
            print(f"ballast {g} GiB: {type(e).__name__}", flush=True)
            break
        sets = [[ev.alloc_mask(P, how=how) for _ in range(R)] for _ in range(K)]
        for m in sets:
            timed(m, 8)
        rows = [timed(m)[0] for m in sets]
        free_b, total_b = torch.cuda.mem_get_info()
        print(f"ballast {g:4d} GiB (free now {free_b / 2**30:6.1f} GiB): candidates at " + " ".join(f"{m[0].data_ptr():#x}" for m in sets) + " : " + " ".join(f"{x:7.2f}" for x in rows) + " us", flush=True)
        del sets, ballast
        torch.cuda.empty_cache()
    rig.close()
    sys.exit(0)

cands = []  # (how, index, [masks])
spacers = []
for hi, how in enumerate(hows):
    if frag:
        spacers.append(fragment(hi))
    for i in range(K):
        try:
            masks = []
            for _ in range(R):
                masks.append(ev.alloc_mask(P, how=how))
                rep = ev.mask_probe_report()
                if rep.size:
                    print(f"how {how} candidate {i}: the library's probe read " + " ".join(f"{x:7.2f}" for x in rep) + f" us per launch (fit only), kept {rep.min():7.2f}", flush=True)
        except L.KschedError as e:
            print(f"how {how} ({L.MASK_ALLOC_NAMES[how]}): allocation failed: {e}", flush=True)
            break
        cands.append((how, i, masks))

res = []
for how, i, masks in cands:
    mean, med, lo = timed(masks)
    res.append((how, i, mean, med, lo, masks[0].data_ptr()))
    print(f"how {how} {L.MASK_ALLOC_NAMES[how]:10s} candidate {i}: first buffer at {masks[0].data_ptr():#x}: mask kernel mean {mean:7.2f} us, median {med:7.2f}, min {lo:7.2f}", flush=True)
PASSES = int(kw.get("passes", 2))
for ps in range(1, PASSES):
    print(f"# pass {ps + 1} (is a candidate's rate its own?)")
    for j, (how, i, masks) in enumerate(cands):
        mean, med, lo = timed(masks)
        print(f"how {how} {L.MASK_ALLOC_NAMES[how]:10s} candidate {i}: mean {mean:7.2f} us (earlier passes " + " ".join(f"{x:7.2f}" for x in res[j][6:] or (res[j][2],)) + ")", flush=True)
        res[j] = res[j] + (mean,) if len(res[j]) > 6 else res[j] + (res[j][2], mean)
for how in hows:
    r = [x for x in res if x[0] == how]
    if r:
        a = np.array([x[6:] if len(x) > 6 else (x[2],) for x in r])
        print(f"== how {how} {L.MASK_ALLOC_NAMES[how]:10s}: mean over candidates and passes {a.mean():7.2f} us, fastest {a.min():7.2f}, slowest {a.max():7.2f}, "
              f"largest change of one candidate between passes {(a.max(axis=1) - a.min(axis=1)).max():6.2f}")
print("SURVEY " + json.dumps({"workload": name, "rows": [{"how": x[0], "name": L.MASK_ALLOC_NAMES[x[0]], "candidate": x[1], "ptr": x[5], "passes_us": list(x[6:] if len(x) > 6 else (x[2],))} for x in res]}))

if mode == "pmc":
    # final phase: n launches per candidate, nothing else on the device; the csv's LAST len(cands) * n mask-kernel dispatches are these
    plan = []
    ev.set_timing(True, every=1)
    for how, i, masks in cands:
        run = bound(masks)
        for k in range(2 * len(masks)):
            run(0, k % len(masks))
        torch.cuda.synchronize()
    ev.kernel_time_samples(8192)
    torch.cuda.synchronize()
    for how, i, masks in cands:
        run = bound(masks)
        for k in range(N):
            run(0, k % len(masks))
        torch.cuda.synchronize()
        us = ev.kernel_time_samples(8192) * 1e3
        plan.append({"how": how, "name": L.MASK_ALLOC_NAMES[how], "candidate": i, "launches": N, "event_mean_us": float(np.mean(us)), "ptr": masks[0].data_ptr()})
    ev.set_timing(False)
    print("PLAN " + json.dumps({"workload": name, "per_candidate_launches": N, "kernel_prefix": "k_eval_fused", "candidates": plan}), flush=True)
del cands
rig.close()
