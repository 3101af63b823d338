#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: pod x node predicate evaluations per second.

    python bench.py --gpus N --steps K --warmup W [--workload C3] [--kernel auto|direct|fused] [--packed]

A "step" is one pass of the hot path over one batch of synthetic pods, inputs already resident in
HBM: the feasibility mask of every (pod, node) pair of this rank's pod rows AND the sampled pick
(select_node_for_pod with injected draws) -- on one GPU that is ONE kernel launch (the pick rides in
the fused mask kernel, KSCHED_OPT_FUSED_PICK) -- and, for N > 1, the all-gather of the int32 bindings
over RCCL.  One evaluation = one (pod, node) feasibility bit.

Workloads (per GPU; weak scaling: rank r evaluates its own P pods against the replicated snapshot):
    C2  10k pods x 1k nodes, fit only                       BASELINE.json configs[1] (launch-bound)
    C3  100k pods x 5k nodes, fit + nodeSelector (8 keys)   BASELINE.json configs[2]  <- default at EVERY N
    C3h C3 with a hostname-like label key (5 000 values)      not a BASELINE config: the high-cardinality case
    C4s 125k pods x 10k nodes, fit + sel                    configs[3] = 8 of these (1M x 10k)
    C5s 125k pods x 50k nodes, fit + sel + taints, best fit configs[4] = 8 of these (1M x 50k)
The default workload is the same at every N (the driver computes scaling from the per-N values: a workload that
changed with N would read as scaling); for N > 1 the line also carries `config.configs3_strong`, the strong-scaling
leg BASELINE.json's configs[3] describes (1M pods x 10k nodes split N ways).

L3-proof: the timed loop rotates its output over enough mask buffers to exceed the 256 MiB Infinity Cache
(`config.mask_rotation`), so a step's stores cannot be absorbed by the cache holding the previous step's mask; the
in-place figure (one buffer rewritten every step) is reported next to it (`config.in_place`).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant (mask) kernel: algorithmic bytes
per launch / its mean HIP-event duration, measured live in this run on the launch stream: events on
every mask kernel dispatch of a post-pass of --kernel-samples further steps behind the same untimed run-in
as the timed region (events cost launch gap, so the timed steps carry none).  `roofline.traffic` is measured
by this invocation too (--live-traffic: separate calibrated rocprofv3 --pmc passes behind the timed region;
the committed profiles/pmc_traffic.json figure stays beside it and is the fallback).
`cpu_baseline` is the oracle (CPU restatement, kind "port") timed on this box's host cores on a
bounded sample of the same workload; it is a reported baseline, never the thing measured above.
`python bench.py --gpus N` without WORLD_SIZE launches its N ranks itself (torch.distributed.run, 127.0.0.1).
Test hook: KSCHED_TEST_HOOKS=1 KSCHED_RCCL_LIB=tests/cpp/libfake_rccl.so KSCHED_BENCH_ONE_GPU=1 puts all N ranks
on device 0 (the N > 1 code end to end on a one-GPU box; the line says it is not a scaling figure).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
HBM_COPY_CEILING_GBS = 6290.0  # measured float4-copy ceiling, same table
L3_BYTES = 256 << 20        # Infinity Cache (MALL), same guide

WORKLOADS = {
    # name: (config, P per GPU, N, flags, pick, description)
    "C2": ("C2", 10_000, 1_000, ("FIT",), "sampled", "C2: 10k pods x 1k nodes, fit only (BASELINE.json configs[1])"),
    "C3": ("C3", 100_000, 5_000, ("FIT", "SEL"), "sampled",
           "C3: 100k pods x 5k nodes, fit + nodeSelector over 8 label keys (BASELINE.json configs[2])"),
    "C3h": ("C3h", 100_000, 5_000, ("FIT", "SEL"), "sampled",
            "C3 with a hostname-like eighth label key (5 000 values, one per node; 15 % of the pods name a node): the high-cardinality case, not a BASELINE config"),
    "C5hs": ("C5h", 125_000, 50_000, ("FIT", "SEL", "TAINT"), "bestfit",
             "C5 shard with a hostname-like eighth label key (50 000 values; 15 % of the pods name a node), best-fit pick: not a BASELINE config"),
    "C4s": ("C4", 125_000, 10_000, ("FIT", "SEL"), "sampled",
            "C4 shard: 125k pods x 10k nodes per GPU, fit + sel (BASELINE.json configs[3] = 8 shards)"),
    "C5s": ("C5", 125_000, 50_000, ("FIT", "SEL", "TAINT"), "bestfit",
            "C5 shard: 125k pods x 50k nodes per GPU, fit + sel + taints, best-fit pick (configs[4] = 8 shards)"),
    "C3x4": ("C3", 400_000, 5_000, ("FIT", "SEL"), "sampled",
             "four C3 batches a caller has queued, passed as ONE call (400k pods x 5k nodes): the tile index is staged once per block for all of them; the tile-test pick rides in the launch (up to 524 288 pods per call)"),
}


def algorithmic_bytes(P, N, n_keys, taint, masks=1, pick_attempts=0):
    """`P*b_pod + N*b_node + P*W*8*m + P*4`, every input column read once, every output written once.
    b_pod = 16 (fit) + 4 per label key + 8 (tolerations) + 4 per draw when the sampled pick runs IN this kernel (`pick_attempts`:
    the launch then also reads the pod's injected draws and writes its int32 binding); without a riding pick the kernel neither
    reads draws nor writes bindings and both terms are left out."""
    W = (N + 63) // 64
    b_pod = 16 + 4 * n_keys + (8 if taint else 0) + 4 * pick_attempts
    b_node = 16 + 4 * n_keys + (8 if taint else 0)
    return P * b_pod + N * b_node + P * W * 8 * masks + (P * 4 if pick_attempts else 0)


def cpu_baseline(c, flags_names, budget_s=12.0, p_cap=None):
    """Time the oracle (object-level C restatement, per-pair string/map evaluation; quantities parsed
    once, LIST once per node) on all host cores on a bounded sample of this workload's pods."""
    from oracle import capi
    flags = sum(getattr(capi, f) for f in flags_names)
    threads = capi.num_threads()
    nodes, bound = c.node_objects(), c.bound_pod_objects()
    s = capi.ObjectSet()
    cn, cb = s.nodes(nodes), s.pods(bound)

    def run(P_s, th):
        pods = c.pod_objects(0, P_s)
        cp = s.pods(pods)
        t0 = time.perf_counter()
        capi.eval_objects(pods, nodes, bound, flags, threads=th, prebuilt=(s, cp, cn, cb))
        return time.perf_counter() - t0

    P_all = min(c.P, p_cap or c.P)  # (the cluster may hold several input batches: the sample stays inside the first)
    probe = min(P_all, 64 * max(1, threads))
    t = run(probe, threads)
    rate = probe * c.N / max(t, 1e-9)
    P_s = int(min(P_all, max(probe, rate * budget_s / c.N)))
    t = run(P_s, threads)
    value = P_s * c.N / t
    # single core, smaller sample
    P_1 = max(1, min(P_s, int(P_s / max(1, threads))))
    t1 = run(P_1, 1)
    # second, stronger CPU baseline: scalar loop on the encoded integer columns
    t0 = time.perf_counter()
    P_e = min(P_all, 20_000)
    capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels if c.n_keys else None, c.node_taints if c.n_taints else None,
                      c.req_cpu[:P_e], c.req_mem[:P_e], c.pod_sel[:, :P_e] if c.n_keys else None,
                      c.pod_tol[:P_e] if c.n_taints else None, None, flags, threads=threads)
    te = time.perf_counter() - t0
    return {
        "value": value, "unit": "evals/s", "cores": threads, "kind": "port",
        "sample": f"{P_s} pods x {c.N} nodes of the same workload, object-level oracle (oracle.c ora_eval_objects: "
                  f"per-pair string/map evaluation, quantities parsed once), {threads} threads, {t:.1f} s",
        "single_core_value": P_1 * c.N / t1,
        "encoded_loop_value": P_e * c.N / te,
        "encoded_loop_note": f"scalar loop on the encoded integer columns (ora_eval_encoded), {threads} threads, {P_e} pods",
        "host_cores": os.cpu_count(),
    }


def end_to_end(torch, L, Evaluator, dev, c, flag_names, pick, objects=True, reps=24):
    """"Separately report end-to-end including H2D/D2H".  Three figures per batch of this workload, none of them the metric's `value`
    (that one is quoted with the inputs resident in HBM):
      host_arrays_to_bindings   numpy columns in pageable host memory -> ksched_eval -> the int32 bindings back in host memory (no mask copy):
                                what the drop-in's reconciler asks for (copies in, ONE launch, 4 bytes per pod out); median of `reps` calls
      host_arrays_to_mask       the same call with the feasibility mask copied back too (63 MB at C3, pageable): what a caller pays that wants the matrix,
                                into result arrays the caller keeps from batch to batch; `fresh_output_ms_per_batch` next to it = into a newly allocated array
                                every call (its pages are first touched by the copy itself: four times slower, and what rounds 5 - 6 quoted)
      objects                   corev1 objects (pods with quantity strings and selector maps) -> reconcile_batch of the C++ host mirror
                                (draws, encode, device, binding POSTs through a recording sink, the snapshot update) -- tests/cpp/objects_eval
                                on a cluster of this workload's shape, best and median of 5 batches each against a fresh snapshot, the WARN level off."""
    flags = sum(getattr(L, f) for f in flag_names) | (L.PICK_SAMPLED if pick == "sampled" else L.PICK_BESTFIT)
    out = {"workload": f"{c.P} pods x {c.N} nodes per batch", "host_cores": os.cpu_count()}
    try:
        ev = Evaluator(dev.index)
        ev.set_nodes(**c.node_columns())
        args = (c.req_cpu, c.req_mem, c.pod_sel if c.n_keys else None, c.pod_tol if "TAINT" in flag_names else None, c.samples if pick == "sampled" else None, flags)
        for key, want_mask, n in (("host_arrays_to_bindings", False, reps), ("host_arrays_to_mask", True, max(5, reps // 3))):
            keep = ev.eval(*args, want_mask=want_mask)  # (scratch allocations of the host-pointer path; the result arrays every later call writes again)
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                ev.eval(*args, want_mask=want_mask, out=keep)
                ts.append(time.perf_counter() - t0)
            med = float(np.median(ts))
            out[key] = {"ms_per_batch": med * 1e3, "min_ms": float(np.min(ts)) * 1e3, "calls": n, "evals_per_s": float(c.P) * c.N / med, "output": "result arrays kept from call to call"}
            if want_mask:
                tf = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    ev.eval(*args, want_mask=True)
                    tf.append(time.perf_counter() - t0)
                out[key]["fresh_output_ms_per_batch"] = float(np.median(tf)) * 1e3
        ev.close()
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    if objects:
        try:
            import importlib.util
            import re
            spec = importlib.util.spec_from_file_location("ksched_host_loop", os.path.join(ROOT, "tools", "host_loop.py"))
            hl = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(hl)
            path = hl.objects_file(c.P, c.N)
            try:
                d, timing, _ = hl.run("batch", path, reps=5)
            finally:
                os.unlink(path)
            secs = sorted(d["seconds_all"])
            split = None
            rows = [re.findall(r"([0-9.]+) ms", ln) for ln in timing if ln.startswith("reconcile_batch")]
            if rows:
                a = np.median(np.array([[float(x) for x in r[:4]] for r in rows if len(r) >= 4]), axis=0)
                split = {"draws_encode_device_ms": float(a[0]), "posts_with_the_update_staged_beside_them_ms": float(a[1]), "snapshot_update_committed_ms": float(a[2]), "warn_lines_ms": float(a[3])}
            out["objects"] = {"what": "corev1 objects -> reconcile_batch (C++ host mirror) -> bindings POSTed to a recording sink -> snapshot updated; per batch, fresh snapshot each",
                              "best_ms_per_batch": secs[0] * 1e3, "median_ms_per_batch": secs[len(secs) // 2] * 1e3, "batches": len(secs), "pods_bound": d["posted_count"],
                              "pods_per_s": c.P / secs[0], "median_split": split}
        except Exception as e:  # noqa: BLE001
            out["objects"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def rotation_for(mask_bytes: int, want: bool) -> int:
    """Mask buffers the timed loop rotates over: enough that their total exceeds the Infinity Cache by a quarter (a step's
    stores then cannot land in lines the cache still holds from the step that last wrote the same buffer)."""
    if not want or mask_bytes <= 0:
        return 1
    return int(max(1, min(12, -(-(L3_BYTES * 5 // 4) // mask_bytes))))


class SingleRig:
    """One workload on one GPU, strictly sequential steps on the current stream (the N = 1 form): used for the secondary figures
    of the default line (`config.other_workloads`, `config.in_place`) -- same code path as the graded loop, its own evaluator."""

    def __init__(self, torch, L, synth, Evaluator, dev, name, kernel="auto", fused_pick=1, packed=False, debug=0):
        cfg, P, N, flag_names, pick, desc = WORKLOADS[name]
        self.torch, self.name, self.desc, self.P, self.N, self.flag_names, self.pick = torch, name, desc, P, N, flag_names, pick
        c = synth.make_config(cfg, P=P, N=N)
        self.c = c
        self.flags = sum(getattr(L, f) for f in flag_names) | (L.PICK_SAMPLED if pick == "sampled" else L.PICK_BESTFIT)
        self.taint = "TAINT" in flag_names
        ev = Evaluator(dev.index)
        ev.set_kernel(kernel)
        ev.set_option(L.OPT_FUSED_PICK, fused_pick)
        if debug:
            ev.set_option(L.OPT_DEBUG, debug)
        ev.set_nodes(**c.node_columns())
        self.ev = ev
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
        self.d = (t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32) if c.n_keys else None,
                  t(c.pod_tol, np.int64) if self.taint else None, t(c.samples, np.int32) if pick == "sampled" else None)
        self.out = torch.full((P,), -1, dtype=torch.int32, device=dev)
        self.packed = packed

    def loop(self, rotate: bool, mask: bool = True):
        ev, torch = self.ev, self.torch
        if not mask:
            run = ev.bind_eval_device(*self.d, self.flags, out_feasible=None, out_bindings=[self.out])
            return (lambda: run(0, 0)), 0, None
        m0 = ev.alloc_mask(self.P, pitched=not self.packed)
        R = rotation_for(int(m0.stride(0)) * 8 * self.P, rotate)
        masks = [m0] + [ev.alloc_mask(self.P, pitched=not self.packed) for _ in range(R - 1)]
        run = ev.bind_eval_device(*self.d, self.flags, out_feasible=masks, out_bindings=[self.out])
        k = [0]

        def step():
            run(0, k[0] % R)
            k[0] += 1
        return step, R, masks

    def measure(self, steps: int, samples: int, rotate: bool = True):
        torch, ev = self.torch, self.ev
        step, R, masks = self.loop(rotate)
        for _ in range(max(8, 2 * R)):
            step()
        torch.cuda.synchronize()
        # untimed run-in, like the graded loop's: bursts until two in a row agree within 2 %, 60 ms at least, 0.6 s at most.  (It was added on the belief
        # that a store-bound launch keeps getting faster for a few hundred ms of sustained load -- the C5 shard's mask kernel read 171 us in one
        # measurement and 141 us in another.  tools/clock_ramp.py and tools/placement_probe.py have since shown the rate to be flat in time from the
        # first milliseconds and the two values to be a property of the mask ALLOCATION, profiles/r05_bimodal_by_allocation.md; the run-in is harmless.)
        t_r = time.perf_counter()
        prev = None
        while True:
            t_b = time.perf_counter()
            for _ in range(32):
                step()
            torch.cuda.synchronize()
            now = time.perf_counter()
            burst = now - t_b
            settled = prev is not None and abs(burst - prev) <= 0.02 * prev
            prev = burst
            if (now - t_r >= 0.06 and settled) or now - t_r >= 0.6:
                break
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ev.set_timing(True, every=1)
        for _ in range(samples):
            step()
        torch.cuda.synchronize()
        ev.kernel_time_ms()
        for _ in range(samples):
            step()
        torch.cuda.synchronize()
        us = np.sort(ev.kernel_time_samples(samples * 2) * 1e3)
        ev.set_timing(False)
        kern, pick_how = ev.last_kernel, ev.last_pick
        del masks
        # the pick alone: a bindings-only request (sampled: k_select_sampled; best fit: the two best-fit stages); no mask kernel runs
        bstep, _, _ = self.loop(False, mask=False)
        for _ in range(8):
            bstep()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            bstep()
        torch.cuda.synchronize()
        pick_us = (time.perf_counter() - t1) / steps * 1e6
        n_keys = self.c.n_keys if "SEL" in self.flag_names else 0
        alg = algorithmic_bytes(self.P, self.N, n_keys, self.taint, pick_attempts=int(self.c.samples.shape[1]) if (pick_how.startswith("fused") and self.pick == "sampled") else 0)
        avg = float(us.mean()) if us.size else 0.0
        return {"workload": self.desc, "value": float(self.P) * self.N * steps / el, "ms_per_step": el / steps * 1e3, "steps": steps,
                "mask_rotation": R, "kernel": kern, "pick": self.pick, "pick_in_mask_launch": pick_how.startswith("fused"),
                "mask_kernel_us": avg, "mask_kernel_median_us": float(np.median(us)) if us.size else None,
                "mask_kernel_frac": (alg / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS) if avg > 0 else None,
                "step_frac": alg / (el / steps) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": alg, "pick_alone_us_per_step": pick_us,
                "pick_alone_note": "a bindings-only request of the same batch (no mask kernel): launch-to-launch time of the pick by itself"}

    def close(self):
        self.ev.close()



def live_traffic(workload, kernel_arg, extra_args, budget_s=240.0):
    """HBM bytes per launch of the mask kernel, measured now (the line used to carry a constant from an earlier session and
    could not notice a traffic regression).  MI355X_MICROARCH.md's recipe: counters in their own passes (--pmc with --kernel-trace only),
    FETCH_SIZE and WRITE_SIZE separately, each trusted only after calibration on kernels of KNOWN byte counts in the same access pattern
    (tools/pmc_calib: 512 MiB flat 16 B / lane reads, 128-byte-segment mask-shaped stores; the guide's gfx950 x 2 for wide reads comes out
    of the calibration).  The passes re-run THIS file with 10 steps of the same workload, kernel and rotation.  -> (record, None) or (None, why)."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    calib = os.path.join(ROOT, "tools", "pmc_calib")
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    if not os.path.exists(calib):
        return None, "tools/pmc_calib not built (make)"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic as PT
    out = tempfile.mkdtemp(prefix="ksched_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", KSCHED_BENCH_TRAFFIC_CHILD="1")
    child = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--kernel", kernel_arg, "--steps", "10", "--warmup", "2", "--ramp-ms", "0",
             "--kernel-samples", "2", "--no-cpu-baseline", "--no-others", "--repeats", "0", "--live-traffic", "off"] + list(extra_args)
    t_end = time.perf_counter() + budget_s
    res = {}
    try:
        for counter, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            for name, cmd in (("calib_" + tag, [calib]), ("run_" + tag, child)):
                left = t_end - time.perf_counter()
                if left < 5:
                    return None, f"time budget of {budget_s:.0f} s spent before pass {name}"
                r = subprocess.run([rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(out, name), "-o", "p", "--"] + cmd,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
                if r.returncode != 0:
                    return None, f"pass {name} exited {r.returncode}: {(r.stderr or r.stdout)[-200:]!r}"
                res[name] = PT.per_kernel(os.path.join(out, name))
        cf, cw = res["calib_fetch"], res["calib_write"]
        f_factor = PT.CALIB_BYTES["calib_read_flat16"] / cf["calib_read_flat16"]["FETCH_SIZE"]
        w_factor = PT.CALIB_BYTES["calib_write_tile128"] / cw["calib_write_tile128"]["WRITE_SIZE"]
        kf = {k: v for k, v in res["run_fetch"].items() if k.startswith("k_eval")}
        kw = {k: v for k, v in res["run_write"].items() if k.startswith("k_eval")}
        if len(kf) != 1 or set(kf) != set(kw):
            return None, f"expected one mask kernel in the passes, saw {sorted(kf)} / {sorted(kw)}"
        k = next(iter(kf))
        fetch_b, write_b = kf[k]["FETCH_SIZE"] * f_factor, kw[k]["WRITE_SIZE"] * w_factor
        return {"kernel": k, "fetch_bytes": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
                "dispatches": [kf[k]["_dispatches"], kw[k]["_dispatches"]], "fetch_bytes_per_count": f_factor, "write_bytes_per_count": w_factor}, None
    except subprocess.TimeoutExpired as e:
        return None, f"a rocprofv3 pass ran into the time budget ({e.timeout:.0f} s left for it)"
    except Exception as e:  # noqa: BLE001  (the measurement is optional: the line falls back to the committed figure and says so)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: C3 per GPU at every N (configs[2], the largest single-GPU configuration; weak scaling)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "direct", "fused"])
    ap.add_argument("--pods", type=int, default=None, help="override pods per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host-side legs: cpu_baseline and the objects leg of config.end_to_end")
    ap.add_argument("--no-mask", action="store_true", help="bindings only (not the graded form)")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="skip the self-check after the timed region (the last timed step's bindings -- all of this rank's pods -- and >= 4096 of its "
                         "mask rows against the oracle's integer loop; the run exits non-zero on a mismatch)")
    ap.add_argument("--allow-unchecked", action="store_true",
                    help="exit 0 even when the self-check itself could not run (e.g. oracle/liboracle.so missing); the default exits 3: an unverified number is not reported as a good one")
    ap.add_argument("--parity-rows", type=int, default=4096, help="mask rows the self-check compares word for word (spread evenly over the rank's pods)")
    ap.add_argument("--packed", action="store_true",
                    help="mask rows packed at W words (default: rows pitched to ksched_mask_pitch(n) = W rounded up to 128 B)")
    ap.add_argument("--debug", type=int, default=0, help="kernel ablation bits (timing experiments; results invalid)")
    ap.add_argument("--fused-pick", type=int, default=1, choices=[0, 1, 2, 3],
                    help="1 (default): the sampled pick rides in the fused mask launch (one kernel per step), in the form the library chooses; "
                         "0: its own launch ahead of it; 2: rides as waves of the fill; 3: rides as tile tests in phase 1 (KSCHED_OPT_FUSED_PICK)")
    ap.add_argument("--no-rotate", action="store_true",
                    help="rewrite ONE mask buffer every step (the Infinity Cache then absorbs part of the stores at C3 / C4s); the default "
                         "rotates over enough buffers to exceed it and reports the in-place figure as config.in_place")
    ap.add_argument("--input-batches", type=int, default=6,
                    help="the timed loop cycles over this many DIFFERENT seeded pod batches resident in HBM (a scheduler never sees the same "
                         "batch twice; one batch evaluated every step would have its 6.8 MB of operands served from L2), like it cycles over "
                         "mask buffers.  Batch 0 is the workload's standard batch; the self-check checks whichever batch the last timed step evaluated")
    ap.add_argument("--no-others", action="store_true",
                    help="N = 1, default workload: skip config.other_workloads (C4s, C5s measured in the same process, a few hundred ms) "
                         "and config.in_place")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="leave this many of the 256 compute units out of every fused mask launch (KSCHED_OPT_GRID_CUS = 256 - K).  For the first run on a real "
                         "multi-GPU node: RCCL's all-gather kernels run beside a mask kernel whose blocks own every CU's LDS; `--gpus 8` against `--gpus 8 --reserve-cus 8` "
                         "is the A/B (VERDICT r5 item 6b).  0 = the whole chip (default)")
    ap.add_argument("--round-order", type=int, default=0, choices=[0, 1, 2],
                    help="KSCHED_OPT_ROUND_ORDER of the graded loop's evaluator: 0 interleaved wave-major (default), 1 blocked (rounds 1 - 5), 2 interleaved chunk-major")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="N = 1: skip config.end_to_end (host arrays -> bindings / mask, and objects -> reconcile_batch through the C++ host mirror: ~25 s)")
    ap.add_argument("--live-traffic", choices=["auto", "on", "off"], default="auto",
                    help="roofline.traffic measured by THIS invocation: after the timed region, four short rocprofv3 --pmc passes (FETCH_SIZE and "
                         "WRITE_SIZE, separately: over tools/pmc_calib -- known byte counts -- and over this command with 10 steps), per launch of the "
                         "mask kernel.  auto (default): at N = 1, when rocprofv3 and tools/pmc_calib exist and this process is not itself being "
                         "profiled; a pass that fails or times out falls back to the committed profiles/pmc_traffic.json (traffic_source says which)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1: skip config.configs3_strong (1M pods x 10k nodes split N ways)")
    ap.add_argument("--repeats", type=int, default=4, help="further timed regions of K steps after the graded one (config.repeat_ms_per_step)")
    ap.add_argument("--depth", type=int, default=None,
                    help="buffer slots in flight (bindings, and masks with two streams).  Default: 1 on one GPU (strictly sequential steps "
                         "on one stream); 2 for N > 1, where the all-gather is asynchronous and overlaps the kernels of the following steps")
    ap.add_argument("--ramp-ms", type=float, default=60.0,
                    help="after the W warm-up steps keep stepping (untimed) until this many ms have passed: the GPU's clocks ramp over "
                         "tens of milliseconds, and a 5-step warm-up of 20 us steps ends long before that (reported as `ramp_steps`)")
    ap.add_argument("--kernel-samples", type=int, default=64,
                    help="post-pass after the timed region: this many further steps with HIP events on EVERY mask kernel dispatch "
                         "(roofline.avg_kernel_us = their mean; min / median reported).  Events cost launch gap, not kernel time, so "
                         "they are kept out of the timed steps")
    ap.add_argument("--gather-every", type=int, default=None,
                    help="N > 1: one all-gather per this many steps (their bindings share a buffer).  Default 1: one per batch -- north_star's step is "
                         "\"evaluate + RCCL allgather of the resulting bindings\" -- and the same run also times one gather per FOUR batches and "
                         "reports it as config.allgather_every_4 (fewer, larger collectives: replicas then see bindings up to four batches late)")
    ap.add_argument("--torch-gather", action="store_true",
                    help="N > 1: all-gather with torch.distributed.all_gather_into_tensor instead of the C ABI's communicator "
                         "(ksched_allgather_bindings); A/B only, the default is the ABI")
    ap.add_argument("--refresh-every", type=int, default=0,
                    help="every this many steps the snapshot changes before the step: ksched_update_nodes with new `available` values for "
                         "--refresh-nodes nodes (what pod watch events do to a live scheduler).  0 = static snapshot (default)")
    ap.add_argument("--refresh-nodes", type=int, default=8)
    ap.add_argument("--overlap-leg", action="store_true",
                    help="N = 1: after the graded loop, time the same K steps with two batches in flight on two streams and report it as "
                         "config.two_batches_in_flight")
    ap.add_argument("--one-stream", action="store_true", help="N > 1: keep pick, all-gather (side stream) and mask kernel off the two-stream pipe")
    ap.add_argument("--split-pipe", action="store_true",
                    help="N > 1: the pipe's split mode (mask kernels on one stream; pick -> all-gather on the other) instead of the default, "
                         "alternate: whole steps (ONE launch when the pick rides) + their all-gather on stream (slot mod 2)")
    ap.add_argument("--two-stream", action="store_true",
                    help="with --depth >= 2: mask kernels on one HIP stream, pick kernels (+ all-gather) on another (ksched_pipe)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly (`python bench.py --gpus 8 ...`): become the launcher -- one process per GPU under torch.distributed.run on
        # this node, the same command line, rendezvous on 127.0.0.1 (the container's hostname may not resolve)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: re-launching as `%s`\n" % (args.gpus, " ".join(cmd)))
        sys.stderr.flush()
        os.execv(sys.executable, cmd)

    import torch
    import torch.distributed as dist

    from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth
    from kube_scheduler_rs_reference_amd.dist import AbiComm, PipelinedScheduler, ShardedScheduler, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} != --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # TEST HOOK (KSCHED_TEST_HOOKS=1 + KSCHED_BENCH_ONE_GPU=1, tests/test_gpu_bench_contract.py): the N ranks of `--gpus N` all sit on device 0 of a
    # one-GPU box, so that the N > 1 code of this file -- row shards with lo > 0, the communicator behind the C ABI with nranks > 1, the gathered
    # table, the cross-rank self-check, the strong-scaling leg -- runs end to end before an 8-GPU node sees it.  RCCL refuses a device twice
    # in one communicator: the library then talks to the TEST-ONLY stand-in ($KSCHED_RCCL_LIB = tests/cpp/libfake_rccl.so, a blocking all-gather
    # through shared memory) and the control group is gloo on host tensors.  The line says so (`config.one_gpu_stand_in`); its value is NOT a scaling figure.
    one_gpu = os.environ.get("KSCHED_BENCH_ONE_GPU") == "1"
    if one_gpu and not (os.environ.get("KSCHED_TEST_HOOKS") == "1" and os.environ.get("KSCHED_RCCL_LIB")):
        raise SystemExit("KSCHED_BENCH_ONE_GPU=1 is a test hook: it needs KSCHED_TEST_HOOKS=1 and KSCHED_RCCL_LIB=<the RCCL stand-in>")
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if one_gpu else dev  # where the control collectives' tensors live (gloo moves host tensors)
    # KSCHED_BENCH_FORCE_DIST=1 (self-test): run the N > 1 code path -- RCCL process group, asynchronous all-gather of the
    # bindings, barrier, MAX all-reduce of the elapsed time -- in a one-rank group on the one GPU there is
    multi = world > 1 or bool(os.environ.get("KSCHED_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    default_workload = args.workload is None
    if default_workload:
        args.workload = "C3"  # at EVERY N: the driver's scaling figures compare like with like
    cfg, P_gpu, N, flag_names, pick, desc = WORKLOADS[args.workload]
    if args.pods:
        P_gpu = args.pods

    def sync():
        # drain this rank's streams first (so the barrier's collective never interleaves with all-gathers still in flight on the
        # ABI's communicator), then barrier + synchronize
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if not multi:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    class Rig:
        """This rank's shard of one P_total x N workload: evaluator, resident inputs, loops."""

        def __init__(self, cfg, P_total, N, flag_names, pick, n_batches=None):
            self.P_total, self.N, self.flag_names, self.pick = P_total, N, flag_names, pick
            # same seeded cluster on every rank; each takes its rows.  B pod batches of P_total pods each: the generator's streams are
            # prefixes of one another, so batch 0 is exactly the P_total-pod workload and batches 1.. are further pods of the same distribution
            self.B = B = max(1, n_batches or args.input_batches)
            self.lo, self.hi, _ = shard_bounds(P_total, world, rank)
            lo, hi = self.lo, self.hi
            # (a rank generates only ITS rows of every batch: pods [b * P_total + lo, b * P_total + hi) of the seed's pod sequence, synth pod_offset;
            # at N = 8 everybody's rows of six batches would be 4.8 M pods and 11 s of host time per rank)
            self.cs = [synth.make_config(cfg, P=hi - lo, N=N, pod_offset=b * P_total + lo) for b in range(B)]
            c = self.cs[0]
            self.c = c
            flags = sum(getattr(L, f) for f in flag_names)
            flags |= L.PICK_SAMPLED if pick == "sampled" else L.PICK_BESTFIT
            self.flags = flags
            self.taint = "TAINT" in flag_names
            ev = Evaluator(local_rank)
            ev.set_kernel(args.kernel)
            ev.set_option(L.OPT_FUSED_PICK, args.fused_pick)
            if args.debug:
                ev.set_option(L.OPT_DEBUG, args.debug)
            if args.reserve_cus:
                ev.set_option(L.OPT_GRID_CUS, 256 - args.reserve_cus)
            if args.round_order:
                ev.set_option(L.OPT_ROUND_ORDER, args.round_order)
            ev.set_nodes(**c.node_columns())
            self.ev = ev
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
            self.batches = []  # per input batch: (req_cpu, req_mem, sel, tol, samples) of THIS rank's rows, resident
            for cb in self.cs:
                self.batches.append((t(cb.req_cpu, np.int64), t(cb.req_mem, np.int64),
                                     t(cb.pod_sel, np.int32) if cb.n_keys else None,
                                     t(cb.pod_tol, np.int64) if self.taint else None,
                                     t(cb.samples, np.int32) if pick == "sampled" else None))
            self.d_cpu, self.d_mem, self.d_sel, self.d_tol, self.d_smp = self.batches[0]
            self.input_bytes = sum(int(x.numel() * x.element_size()) for bt in self.batches for x in bt if x is not None)
            self.comm, self.comm_note = None, None

        def make_comm(self):
            """ksched_comm_create: the C ABI's RCCL communicator.  Its creation is collective; if it fails on ANY rank (e.g. no
            usable librccl for dlopen) every rank falls back to torch's collective together, and the JSON line says so."""
            if not multi or args.torch_gather:
                return
            try:
                self.comm = AbiComm(self.ev)
                ok = 1
            except Exception as e:  # noqa: BLE001
                self.comm, ok, self.comm_note = None, 0, f"AbiComm failed on rank {rank}: {e}"
            okt = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()) == 0:
                if self.comm is not None:
                    self.comm.close()
                self.comm = None
                self.comm_note = self.comm_note or "AbiComm failed on another rank"
                args.torch_gather = True

        def close(self):
            if self.comm is not None:
                self.comm.close()
            self.ev.close()

    rig = Rig(cfg, P_gpu * world, N, flag_names, pick)
    rig.make_comm()
    ev, c, flags, taint = rig.ev, rig.c, rig.flags, rig.taint
    P_total, lo, hi = rig.P_total, rig.lo, rig.hi
    if args.depth:
        depth = args.depth
    elif multi:
        # N > 1: the pipe owns one mask per slot, so the slots ARE the rotation: as many (an even number: a slot keeps its stream) as it takes for the
        # masks in flight to exceed the Infinity Cache by a quarter, like the N = 1 loop's rotation ("the headline config lives inside the
        # Infinity Cache"; two slots of 61 MiB would)
        w_words = (N + 63) // 64
        pitch_words = w_words if args.packed else (w_words + 15) // 16 * 16
        r_need = 1 if (args.no_mask or args.no_rotate) else rotation_for(pitch_words * 8 * P_gpu, True)
        depth = max(2, r_need + (r_need & 1))
    else:
        depth = 1
    pipelined = depth > 1 and not args.no_mask
    # N > 1: ONE RCCL all-gather of the bindings per batch (north_star: "evaluate + RCCL allgather of the resulting bindings"; VERDICT r3:
    # the amortised form left the exchange out of 3 of 4 timed steps).  The pipe runs in its alternate mode: batch i = one launch (mask +
    # riding pick) on stream (i mod 2) and its all-gather right behind it on the same stream, so the gather of batch i overlaps the launch
    # of batch i + 1 on the other stream.  --gather-every G > 1 = the bindings of G consecutive batches in one collective; the default run
    # also times G = 4 and reports it next to the primary number (config.allgather_every_4).
    gather_every = max(1, args.gather_every or 1) if (multi and pipelined) else 1
    if multi and pipelined and not args.one_stream:
        # N > 1 default: ksched_pipe -- mask kernels on one stream; pick -> all-gather -> pick -> ... on the other.  The gather is
        # ordered behind its pick by the stream itself (no event per step) and overlaps the next batches' mask kernels.
        args.two_stream = True
    W = ev.W

    class Loop:
        """One configuration of the step loop: scheduler (sharding + gather), buffers, pre-marshalled launches."""

        def __init__(self, rig, G, depth=depth, two_stream=None, rotate=True, alternate=False, inputs_rotate=True):
            ev = rig.ev
            self.rig = rig
            self.alternate = alternate and G == 1 and depth % 2 == 0
            n_loc = rig.hi - rig.lo
            pipelined = depth > 1 and not args.no_mask
            two_stream = args.two_stream if two_stream is None else two_stream
            self.G, self.depth, self.pipelined = G, depth, pipelined
            self.pipe = ev.pipe(depth * G) if (pipelined and two_stream) else None
            self.alternate = self.alternate and self.pipe is not None
            self.ev = ev
            self.use()
            self.sched = (PipelinedScheduler(rig.P_total, dev, depth=depth, pipe=self.pipe, gather_always=multi, gather_every=G, comm=rig.comm,
                                             alternate=self.alternate)
                          if pipelined else ShardedScheduler(rig.P_total, dev, comm=rig.comm))
            sched = self.sched
            probe = None if args.no_mask else ev.alloc_mask(n_loc, pitched=not args.packed)
            mask_bytes = 0 if probe is None else int(probe.stride(0) if n_loc > 1 else ev.W) * 8 * n_loc
            # rotation: the sequential loop cycles its output over R buffers (> 256 MiB together); the pipe already owns
            # depth * G masks and takes further ones until the same total is reached
            want_rot = rotate and not args.no_rotate and not args.no_mask
            R = rotation_for(mask_bytes, want_rot)
            if self.pipe is not None:
                n_masks = depth * G
                self.R = n_masks
            else:
                n_masks = R
                self.R = R
            self.masks = [None] if args.no_mask else [probe] + [ev.alloc_mask(n_loc, pitched=not args.packed) for _ in range(n_masks - 1)]
            self.mask_bytes = mask_bytes
            if pipelined:  # one binding buffer per (slot, step of the slot's gather group)
                slot_outs = {(k, g): sched.binding_buffer(k, g) for k in range(depth) for g in range(G)}
            else:
                slot_outs = {(0, 0): sched.local[: n_loc]}
            keys = sorted(slot_outs)
            # sequential form: ONE (pre-marshalled) library call per step on one stream
            # one pre-marshalled call per INPUT batch (the loop cycles over rig.B resident pod batches as it cycles over mask buffers)
            nb = rig.B if inputs_rotate else 1
            self.n_inputs = nb
            bounds = [ev.bind_eval_device(*rig.batches[b], rig.flags,
                                          out_feasible=None if args.no_mask else (self.masks if self.pipe is None else self.masks[0]),
                                          out_bindings=[slot_outs[k] for k in keys]) for b in range(nb)]
            index_of = {slot_outs[k].data_ptr(): i for i, k in enumerate(keys)}
            submits = None
            if self.pipe is not None:  # pre-marshalled ksched_pipe_submit: mask kernel -> the pipe's mask stream, pick -> its pick stream
                submits = [self.pipe.bind(*rig.batches[b], rig.flags, self.masks, [slot_outs[k] for k in keys]) for b in range(nb)]  # pipe slot = k * G + g
            n_rot = max(1, self.R) if self.pipe is None and not args.no_mask else 1
            k_rot = [0]
            self.last_mask = 0   # index into self.masks of the buffer the latest step wrote (the self-check reads it back)
            self.last_batch = 0  # ... and which input batch it evaluated

            one_out = len(keys) == 1  # (the sequential form has ONE binding buffer: no address lookup per step)

            def local_eval(binding_out):
                k = k_rot[0]
                self.last_mask, self.last_batch = k % n_rot, k % nb
                bounds[self.last_batch](0 if one_out else index_of[binding_out.data_ptr()], self.last_mask)
                k_rot[0] = k + 1

            def run(slot, binding_out):  # pipelined form: bindings and masks are per slot
                self.last_batch = k_rot[0] % nb
                if submits is not None:
                    self.last_mask = slot
                    submits[self.last_batch](slot)
                else:
                    self.last_mask = k_rot[0] % n_rot
                    bounds[self.last_batch](index_of[binding_out.data_ptr()], self.last_mask)
                k_rot[0] += 1
            self.submits = submits
            self.step = (lambda: sched.step(run)) if pipelined else (lambda: sched.step(local_eval))

        def reference(self, b):
            """bindings of input batch b by ONE sequential library call on the current stream (what the other legs' last steps are compared with)"""
            n_loc = self.rig.hi - self.rig.lo
            out = torch.full((n_loc,), -3, dtype=torch.int32, device=dev)
            self.drain()
            torch.cuda.synchronize()
            self.ev.bind_eval_device(*self.rig.batches[b], self.rig.flags, out_feasible=None if args.no_mask else self.masks[0], out_bindings=[out])(0, 0)
            torch.cuda.synchronize()
            return out

        def use(self):
            """KSCHED_OPT_PIPE_MODE is a property of the evaluator, read by every ksched_pipe_submit: set it to this loop's mode before
            this loop's steps are issued (several loops share one evaluator, one at a time)"""
            if self.pipe is not None:
                self.ev.set_option(L.OPT_PIPE_MODE, 1 if self.alternate else 0)

        def drain(self):
            if self.pipelined:
                self.sched.drain()

        def timed(self, steps):
            """exactly `steps` steps between barrier + synchronize pairs; MAX over ranks"""
            self.use()
            sync()
            t0 = time.perf_counter()
            last = None
            for _ in range(steps):
                last = self.step()
            self.drain()
            sync()
            return max_over_ranks(time.perf_counter() - t0), last

        def close(self):
            if self.pipe is not None:
                self.pipe.close()

    # N > 1 default: whole steps alternate between the pipe's two streams, each step's all-gather behind it on its own stream
    alt_default = multi and pipelined and not args.split_pipe and not args.one_stream and gather_every == 1 and depth % 2 == 0
    loop = Loop(rig, gather_every, alternate=alt_default)
    sched, pipe = loop.sched, loop.pipe
    d_mask = loop.masks[0]
    pitch = int(d_mask.stride(0)) if d_mask is not None else W

    refresh = {"n": 0, "calls": 0}
    if args.refresh_every > 0:
        rr = np.random.default_rng(1234 + rank)
        r_idx = [rr.choice(N, size=min(N, args.refresh_nodes), replace=False).astype(np.uint32) for _ in range(64)]
        r_cpu = [c.avail_cpu[i] - rr.integers(0, 500, i.size) for i in r_idx]
        r_mem = [c.avail_mem[i] - rr.integers(0, 1 << 28, i.size) for i in r_idx]

    def one_step():
        if args.refresh_every > 0:
            refresh["n"] += 1
            if refresh["n"] % args.refresh_every == 0:
                j = refresh["calls"] % 64
                ev.update_nodes(r_idx[j], r_cpu[j], r_mem[j])  # enqueued on the ctx's stream; the next evaluation is ordered behind it by an event
                refresh["calls"] += 1
        return loop.step()

    if args.refresh_every > 0:
        # The HIP runtime stalls ONCE per process for ~37 ms around its ~600th snapshot update (profiles/r02_m_snapshot_stream.txt);
        # get past that point before anything is timed, or it lands in the timed region of some --refresh-every / --steps combinations.
        for j in range(700):
            ev.update_nodes(r_idx[j % 64], r_cpu[j % 64], r_mem[j % 64])
        ev.set_nodes(**c.node_columns())
        torch.cuda.synchronize()
    loop.use()
    for _ in range(args.warmup):
        last = one_step()

    # clock ramp (untimed, reported as `ramp_steps`): step until about --ramp-ms have passed.  The step count is agreed across
    # ranks (MAX) so that every rank issues the same collectives.
    def burst(k):
        t_b = time.perf_counter()
        for _ in range(k):
            one_step()
        if pipelined:
            sched.drain()
        torch.cuda.synchronize()
        return time.perf_counter() - t_b
    t16 = burst(16)
    more = int(min(50_000, max(0.0, args.ramp_ms * 1e-3 - t16) / max(t16 / 16, 1e-7)))
    if multi:
        mt = torch.tensor([more], dtype=torch.int64, device=cdev)
        dist.all_reduce(mt, op=dist.ReduceOp.MAX)
        more = int(mt.item())
    if more:
        burst(more)
    ramp_steps = 16 + more
    sync()
    # ---- the timed region: exactly K steps, no events inside -------------------------------------------------------
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one_step()
    t_issued = time.perf_counter()  # (read inside the region: 30 ns; says how the region splits into the host's launch loop and the closing wait)
    if pipelined:
        sched.drain()
    sync()
    t_done = time.perf_counter()
    elapsed = max_over_ranks(t_done - t0)
    region_split = {"launch_loop_us": (t_issued - t0) * 1e6, "closing_wait_us": (t_done - t_issued) * 1e6}
    bindings = (last.wait() if pipelined else last).clone()
    # ---- self-check ("the driver-run bench asserts nothing"): what the LAST TIMED STEP wrote -- this rank's bindings, all of
    # them, and --parity-rows of its mask rows word for word -- against the oracle's scalar loop on the encoded columns (test
    # infrastructure, used here as the checker only; it never runs inside a timed region).  A mismatch is in the JSON line AND the exit code.
    parity = None
    if not args.no_parity_check:
        try:
            from oracle import capi
            torch.cuda.synchronize()
            n_loc = hi - lo
            cb = rig.cs[loop.last_batch]  # the input batch the last timed step evaluated (this rank's rows of it)
            b_cpu, b_mem, b_tol, b_smp = cb.req_cpu, cb.req_mem, cb.pod_tol, cb.samples
            sel_all = cb.pod_sel if (c.n_keys and "SEL" in flag_names) else None
            o_flags = sum(getattr(capi, f) for f in flag_names) | (capi.PICK_SAMPLED if pick == "sampled" else capi.PICK_BESTFIT)
            oracle_args = dict(avail_cpu=c.avail_cpu, avail_mem=c.avail_mem, label_ids=c.node_labels if sel_all is not None else None,
                               taints=c.node_taints if taint else None)
            # bindings: every pod of the shard when the host has the cores for it, else the first pods of the shard
            threads = capi.num_threads()
            n_b = n_loc if threads >= 8 else min(n_loc, max(1, int(2e9 / max(N, 1))))
            _, _, want_b = capi.eval_encoded(**oracle_args, req_cpu=b_cpu[:n_b], req_mem=b_mem[:n_b],
                                             sel_ids=None if sel_all is None else np.ascontiguousarray(sel_all[:, :n_b]), tolerations=b_tol[:n_b] if taint else None,
                                             samples=b_smp[:n_b] if pick == "sampled" else None, flags=o_flags, want_mask=False)
            got_b = bindings[lo:lo + n_b].cpu().numpy() if bindings.numel() >= hi else bindings[:n_b].cpu().numpy()
            bad_b = int((got_b != want_b).sum())
            bad_w = rows_checked = words = 0
            if not args.no_mask and n_loc > 0:
                rows = np.unique(np.linspace(0, n_loc - 1, num=min(n_loc, max(1, args.parity_rows))).astype(np.int64))
                want_m, _, _ = capi.eval_encoded(**oracle_args, req_cpu=b_cpu[rows], req_mem=b_mem[rows],
                                                 sel_ids=None if sel_all is None else np.ascontiguousarray(sel_all[:, rows]),
                                                 tolerations=b_tol[rows] if taint else None, samples=None,
                                                 flags=o_flags & ~(capi.PICK_SAMPLED | capi.PICK_BESTFIT), want_mask=True)
                got_m = loop.masks[loop.last_mask][torch.from_numpy(rows).to(dev)].cpu().numpy().view(np.uint64)
                bad_w = int((got_m != want_m).sum())
                rows_checked, words = int(rows.size), int(want_m.size)
            parity = {"checked": "the last timed step's outputs against oracle.c ora_eval_encoded (scalar loop on the encoded columns)",
                      "input_batch": int(loop.last_batch), "bindings": int(n_b), "bindings_of": int(n_loc), "partial": bool(n_b < n_loc),
                      "binding_mismatches": bad_b, "rows": rows_checked, "words": words, "word_mismatches": bad_w,
                      "mismatches": bad_b + bad_w}
        except Exception as e:  # noqa: BLE001 -- the CHECKER could not run (e.g. oracle/liboracle.so missing on this box): said in the line, not a mismatch
            parity = {"error": f"{type(e).__name__}: {e}", "mismatches": None}
        if multi:  # every rank checks its own shard; the line carries the sum (a rank whose checker could not run counts as unchecked, not as a mismatch)
            pm = torch.tensor([parity["mismatches"] or 0, 0 if parity["mismatches"] is not None else 1], dtype=torch.int64, device=cdev)
            dist.all_reduce(pm, op=dist.ReduceOp.SUM)
            parity["mismatches_all_ranks"], parity["ranks_unchecked"] = int(pm[0].item()), int(pm[1].item())
            # ... and every rank must hold the SAME gathered table (each checks its own rows against the oracle above; the other ranks' rows it only
            # has from the all-gather): an order-sensitive sum per shard of the table, the vectors of all ranks compared -- a checksum of checksums
            if bindings.numel() >= P_total and P_total > 0:
                wts = (torch.arange(P_total, device=dev, dtype=torch.int64) % 1000003) + 1
                prod = (bindings[:P_total].to(torch.int64) + 2) * wts
                mine = torch.stack([prod[slice(*shard_bounds(P_total, world, r)[:2])].sum() for r in range(world)]).to(cdev)
                everyone = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(everyone, mine)
                diffs = sum(int((h != everyone[0]).sum().item()) for h in everyone)
                parity["gathered_table"] = {"rows": int(P_total), "shards_hashed": world, "shard_sums_differing_between_ranks": diffs}
                if diffs:
                    parity["mismatches_all_ranks"] += diffs
    # further regions of the same K steps: how much one region of K steps moves from run to run (the graded one is the first)
    repeats = []
    for _ in range(max(0, args.repeats)):
        sync()
        t_r = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        if pipelined:
            sched.drain()
        sync()
        repeats.append(max_over_ranks(time.perf_counter() - t_r) / args.steps * 1e3)
    # ---- post-pass: the same steps with HIP events on every mask kernel dispatch (per-kernel roofline number) ----------

    def kernel_events(step_fn, drain_fn, ev_, n):
        ev_.set_timing(True, every=1)
        for _ in range(max(1, n)):  # first round: creates the event pool (host work), discarded
            step_fn()
        drain_fn()
        sync()
        ev_.kernel_time_ms()  # reset
        for _ in range(max(1, n)):
            step_fn()
        drain_fn()
        sync()
        us = np.sort(ev_.kernel_time_samples(max(1, n) * 2) * 1e3)
        ev_.set_timing(False)
        return us
    # (the same untimed run-in as before the timed region: by now the GPU has sat idle through the self-check's second or so of oracle work on the
    # host, and at --steps 20 the repeats above are 2 ms of load -- the post-pass then read 1 - 2 us per kernel more than the same kernels in the
    # timed region, session r5v; the kernels are to be measured in the conditions of the region they stand for)
    if ramp_steps > 16:
        burst(ramp_steps - 16)
    samples_us = kernel_events(one_step, loop.drain, ev, args.kernel_samples)
    launches = int(samples_us.shape[0])
    kern_ms = float(samples_us.sum()) * 1e-3
    kernel_name, pick_how = ev.last_kernel, ev.last_pick

    # ---- N = 1: the in-place figure next to the rotated one (ONE mask buffer rewritten every step: what round 2 timed) -------
    in_place = None
    if not multi and not pipelined and not args.no_mask and loop.R > 1 and not args.no_others:
        try:
            lp = Loop(rig, 1, rotate=False)
            k_ip = min(args.steps, 100)  # (bounded, like the two-stream leg below: a rocprofv3 summary of this command averages these launches with the graded ones)
            for _ in range(16):
                lp.step()
            e_ip, _ = lp.timed(k_ip)
            us_ip = kernel_events(lp.step, lp.drain, ev, min(16, args.kernel_samples))
            alg_ip = algorithmic_bytes(hi - lo, N, c.n_keys if "SEL" in flag_names else 0, taint,
                                       pick_attempts=int(c.samples.shape[1]) if (ev.last_pick.startswith("fused") and pick == "sampled") else 0)
            in_place = {"value": float(P_total) * N * k_ip / e_ip, "ms_per_step": e_ip / k_ip * 1e3, "steps": k_ip,
                        "mask_kernel_us": float(np.median(us_ip)), "mask_kernel_frac": alg_ip / (float(np.median(us_ip)) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "note": f"one {lp.mask_bytes / 2**20:.0f} MiB mask buffer rewritten every step: the 256 MiB Infinity Cache still holds "
                                "the previous step's lines, and write counters count fabric requests -- not the L3-proof figure"}
            lp.close()
        except Exception as e:  # noqa: BLE001 -- a secondary figure must never take the graded line down with it
            in_place = {"error": f"{type(e).__name__}: {e}"}

    # ---- N > 1, second number: the same K steps with the bindings of FOUR consecutive batches in ONE all-gather (the pipe in its split
    # mode: mask kernels on one stream; pick -> ... -> all-gather on the other, ordered by the stream itself) ------------
    alt = None
    if multi and pipelined and args.gather_every is None:
        loop.drain()
        loop_alt = Loop(rig, 4, alternate=False)
        for _ in range(32):
            loop_alt.step()
        loop_alt.drain()
        e_alt, _ = loop_alt.timed(args.steps)
        alt = {"value": float(P_total) * N * args.steps / e_alt, "ms_per_step": e_alt / args.steps * 1e3, "steps": args.steps, "steps_per_allgather": 4,
               "note": "fewer, larger collectives: every binding still reaches every rank, the replicas' snapshot generation advances every four batches"}
        loop_alt.close()
        loop.use()

    # ---- N = 1, second number: the same K steps with consecutive batches alternating between TWO streams (ksched_pipe in its
    # "alternate" mode: each step is still ONE launch; the next launch's blocks fill while the previous one's last blocks store).
    # The primary number stays the strictly sequential loop: there the kernel has the chip to itself, and its duration -- the
    # roofline figure -- is the one rocprofv3 reports for the same command.
    overlapped = None
    if not multi and not pipelined and not args.no_mask and args.refresh_every == 0 and (args.overlap_leg or (default_workload and not args.no_others)):
        try:
            loop.drain()
            d_ov = max(2, loop.R + (loop.R & 1))  # one mask per slot: the slots rotate over the same > 256 MiB; even, so a slot keeps its stream
            loop_ov = Loop(rig, 1, depth=d_ov, two_stream=True, alternate=True)
            k_ov = min(args.steps, 64)  # (bounded: the leg's launches share the kernel's name in a rocprofv3 summary of this command, and take twice as long each)
            for _ in range(8):
                loop_ov.step()
            loop_ov.drain()
            e_ov, last_ov = loop_ov.timed(k_ov)
            same = bool(torch.equal(last_ov.wait(), loop.reference(loop_ov.last_batch)))  # the same input batch through ONE sequential call
            alg_ov = algorithmic_bytes(hi - lo, N, c.n_keys if "SEL" in flag_names else 0, taint,
                                       pick_attempts=int(c.samples.shape[1]) if (ev.last_pick.startswith("fused") and pick == "sampled") else 0)
            overlapped = {"value": float(P_total) * N * k_ov / e_ov, "ms_per_step": e_ov / k_ov * 1e3, "steps": k_ov,
                          "streams": 2, "mask_buffers": d_ov, "pick_launch": ev.last_pick, "bindings_equal_sequential": same,
                          "step_frac_of_hbm_peak": alg_ov / (e_ov / k_ov) / 1e9 / HBM_PEAK_GBS,
                          "note": "consecutive batches alternate between two HIP streams (KSCHED_OPT_PIPE_MODE = 1): the fill of launch i+1 "
                                  "overlaps the drain of launch i; per-launch durations are longer here, so this is not the roofline leg"}
            loop_ov.close()
            loop.use()
        except Exception as e:  # noqa: BLE001 -- a secondary figure must never take the graded line down with it
            overlapped = {"error": f"{type(e).__name__}: {e}"}

    # ---- N > 1, reference point: this rank's own shard with NO exchange (same kernels, same pipe, no collective), K steps ------
    # per-GPU rate of the same workload without the all-gather: value / (N x min over ranks of this) is the cost of the exchange
    def solo_rate(lp, rg):
        if lp.pipe is None:
            return None
        lp.drain()
        lp.use()
        outs = [lp.sched.binding_buffer(k, g) for k in range(lp.depth) for g in range(lp.G)]
        subs = [lp.pipe.bind(*rg.batches[b], rg.flags, lp.masks, outs) for b in range(lp.n_inputs)]
        nslot, nb_ = lp.depth * lp.G, lp.n_inputs
        for j in range(32):
            subs[j % nb_](j % nslot)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for j in range(args.steps):
            subs[j % nb_](j % nslot)
        torch.cuda.synchronize()
        e_solo = max_over_ranks(time.perf_counter() - t2)
        return {"per_gpu_value_max_time": float(rg.hi - rg.lo) * rg.N * args.steps / e_solo, "ms_per_step": e_solo / args.steps * 1e3,
                "note": "each rank alone: the same ksched_pipe steps on its shard, no all-gather; slowest rank"}
    solo = solo_rate(loop, rig) if (multi and pipelined) else None

    # ---- N > 1: the strong-scaling leg BASELINE.json's configs[3] describes: 1M pods x 10k nodes split N ways ----------------
    strong = None
    if multi and pipelined and default_workload and not args.no_strong_leg:
        try:
            loop.drain()
            cfg4, _, N4, fn4, pick4, _ = WORKLOADS["C4s"]
            rig4 = Rig(cfg4, 1_000_000, N4, fn4, pick4, n_batches=1)  # (a million pods per batch: one resident batch)
            rig4.comm = rig.comm  # one communicator per process is enough (same ranks, same device)
            lp4 = Loop(rig4, 1, alternate=alt_default)
            for _ in range(16):
                lp4.step()
            lp4.drain()
            k4 = max(20, min(args.steps, 400))
            e4, _ = lp4.timed(k4)
            s4 = solo_rate(lp4, rig4)
            strong = {"workload": f"configs[3]: 1M pods x 10k nodes, pod rows split {world} ways ({rig4.hi - rig4.lo} pods on this rank), "
                                  "one all-gather of the 1M bindings per batch", "scaling": "strong", "pods_total": 1_000_000, "nodes": N4,
                      "value": 1e6 * N4 * k4 / e4, "ms_per_step": e4 / k4 * 1e3, "steps": k4, "no_allgather": s4}
            lp4.close()
            rig4.comm = None
            rig4.close()
        except Exception as e:  # noqa: BLE001
            strong = {"error": f"{type(e).__name__}: {e}"}

    # ---- N = 1, default workload: C4s and C5s in the same process (their own evaluators; the graded loop is over) -------------
    others = None
    if not multi and default_workload and not args.no_others and not args.no_mask and args.refresh_every == 0 and not args.pods:
        others = {}
        for name in ("C4s", "C5s", "C3x4"):
            try:
                r = SingleRig(torch, L, synth, Evaluator, dev, name, kernel=args.kernel, fused_pick=args.fused_pick, packed=args.packed, debug=args.debug)
                others[name] = r.measure(steps=100 if name == "C5s" else 200, samples=24)
                if name == "C3x4":
                    others[name]["us_per_100k_pods"] = others[name]["ms_per_step"] * 1e3 / 4.0
                r.close()
                del r
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                others[name] = {"error": f"{type(e).__name__}: {e}"}

    # ---- N = 1: the end-to-end figures SURVEY.md 8d asks for, beside (never instead of) the resident-input metric ------------------
    e2e = None
    if not multi and not args.no_end_to_end and not args.no_mask and not args.pods and not args.debug:
        # (the objects leg -- 25 s: a cluster's worth of JSON is made and parsed -- with the default workload's full line only: --no-cpu-baseline, the
        # switch of every quick run, skips it like it skips the other host-side leg; the profiler's child runs skip everything)
        if not os.environ.get("KSCHED_BENCH_TRAFFIC_CHILD"):
            e2e = end_to_end(torch, L, Evaluator, dev, c, flag_names, pick, objects=default_workload and not args.no_cpu_baseline)

    # sanity inside the bench: the fraction of pods the last timed step bound (a degenerate workload would show 0 or 1)
    bound_frac = float((bindings >= 0).float().mean().item())

    if rank == 0:
        evals = float(P_total) * N * args.steps
        value = evals / elapsed
        avg_kernel_s = (kern_ms / max(launches, 1)) * 1e-3
        attempts_in_kernel = int(c.samples.shape[1]) if (pick_how.startswith("fused") and pick == "sampled") else 0
        alg = algorithmic_bytes(hi - lo, N, c.n_keys if "SEL" in flag_names else 0, taint, pick_attempts=attempts_in_kernel)
        alg_mask_only = algorithmic_bytes(hi - lo, N, c.n_keys if "SEL" in flag_names else 0, taint)
        achieved = alg / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        # HBM traffic per launch comes from SEPARATE rocprofv3 --pmc passes (tools/gpu_pmc.sh -> tools/pmc_traffic.py), not from
        # this run: the counters cannot be collected inside a timing run.  The file names the session it was measured in.
        traffic, traffic_source, traffic_committed, traffic_live = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                doc = json.load(open(tpath))
                rec = doc.get(f"{args.workload}:{kernel_name}")
                traffic = traffic_committed = rec["hbm_bytes_per_launch"] if rec else None
                if rec:
                    traffic_source = f"profiles/pmc_traffic.json ({doc.get('_session', 'session unnamed')}): separate rocprofv3 --pmc passes, not this run"
            except Exception:
                traffic = None
        # ... and measured by this invocation where that is possible (--live-traffic): the committed figure stays beside it for comparison
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
        try:  # (the profiler's tool library mapped into this process says it most reliably)
            profiled = profiled or any("rocprofiler-sdk-tool" in ln or "librocprofv3" in ln for ln in open("/proc/self/maps"))
        except OSError:
            pass
        want_live = args.live_traffic == "on" or (args.live_traffic == "auto" and world == 1 and not profiled and kernel_name in ("fused", "direct")
                                                   and not args.no_mask and not args.debug and not os.environ.get("KSCHED_BENCH_FORCE_DIST"))
        if want_live and world == 1:
            passthrough = []
            for flag, on in (("--no-rotate", args.no_rotate), ("--packed", args.packed)):
                if on:
                    passthrough.append(flag)
            passthrough += ["--fused-pick", str(args.fused_pick), "--input-batches", str(args.input_batches)]
            if args.pods:
                passthrough += ["--pods", str(args.pods)]
            rec_live, why = live_traffic(args.workload, args.kernel, passthrough)
            if rec_live and kernel_name and rec_live["kernel"].replace("k_eval_", "") == kernel_name:
                traffic_live = rec_live
                traffic = rec_live["hbm_bytes_per_launch"]
                traffic_source = ("this invocation: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this command (10 steps, %d dispatches) "
                                  "and over tools/pmc_calib (known byte counts)" % rec_live["dispatches"][0])
            else:
                traffic_live = {"error": why or f"the passes saw kernel {rec_live['kernel']}, the run timed {kernel_name}"}
                if traffic_source:
                    traffic_source += "; the live passes failed: " + traffic_live["error"]
        step_s = elapsed / args.steps
        per_gpu_solo = solo["per_gpu_value_max_time"] if solo else None
        out = {
            "metric": "pod x node predicate evals/s", "value": value, "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ramp_steps": ramp_steps,
            "untimed_steps_before_timed_region": args.warmup + ramp_steps, "ms_per_step": step_s * 1e3, "timed_region_split": region_split,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": desc, "pods_per_gpu": P_gpu, "pods_total": P_total, "nodes": N,
                       "workload_note": ("the same per-GPU workload at every N (weak scaling): the driver derives scaling efficiency from the per-N values, which a workload that "
                                         "changed with N would corrupt -- BASELINE.json's 8-GPU configuration, configs[3] = 1M pods x 10k nodes split N ways, is measured in the same "
                                         "run as config.configs3_strong" if world > 1 else None),
                       "grid_cus": (256 - args.reserve_cus) if args.reserve_cus else 256, "round_order": args.round_order,
                       "mask_allocation": (None if args.no_mask else ("packed torch tensors" if args.packed else
                                           "ksched_mask_alloc(KSCHED_MASK_ALLOC_AUTO): probe-and-keep for masks >= 128 MiB (config.other_workloads' C4s / C5s), plain hipMalloc below (this workload's "
                                           f"{loop.mask_bytes / 2**20:.0f} MiB masks)" if loop.mask_bytes < (128 << 20) else "ksched_mask_alloc(KSCHED_MASK_ALLOC_AUTO): probe-and-keep (masks >= 128 MiB)")),
                       **({"one_gpu_stand_in": "TEST HOOK: the %d ranks share ONE GPU and gather through the RCCL stand-in; a check of the N > 1 code, not a scaling figure" % world}
                          if one_gpu else {}),
                       "predicates": "+".join(flag_names), "pick": pick, "mask_written": not args.no_mask,
                       "mask_row_pitch_words": pitch, "mask_words": W,
                       "kernel": kernel_name, "pick_launch": pick_how,
                       "kernels_per_step": (1 if pick_how.startswith("fused") else None),
                       "mask_rotation": loop.R, "mask_rotation_bytes": loop.R * loop.mask_bytes,
                       "mask_rotation_note": (f"the loop writes {loop.R} mask buffers of {loop.mask_bytes / 2**20:.0f} MiB in turn "
                                              f"({loop.R * loop.mask_bytes / 2**20:.0f} MiB > the 256 MiB Infinity Cache)" if loop.R > 1 else
                                              "one mask buffer" + (" (already larger than the 256 MiB Infinity Cache)" if loop.mask_bytes > L3_BYTES else "")),
                       "input_rotation": {"batches": loop.n_inputs, "bytes_resident": rig.input_bytes,
                                          "note": f"the loop cycles over {loop.n_inputs} different seeded pod batches resident in HBM ({rig.input_bytes / 1e6:.1f} MB of operand "
                                                  "columns: more than the 8 x 4 MiB of L2), batch 0 = the workload's standard batch; the self-check reads the batch the last "
                                                  "timed step evaluated (parity_check.input_batch)"},
                       "step_frac_of_hbm_peak": (alg / step_s / 1e9 / HBM_PEAK_GBS) if world == 1 else None,
                       "step_frac_note": "algorithmic bytes of one step (mask + draws + bindings) / ms_per_step / 8 TB/s",
                       "repeat_ms_per_step": repeats, "in_place": in_place, "other_workloads": others, "end_to_end": e2e,
                       "steps_in_flight": depth if pipelined else 1, "two_stream": pipe is not None,
                       "pipe_mode": (("alternate: whole steps + their all-gather on stream (slot mod 2)" if loop.alternate else
                                      "split: mask kernels on one stream, pick -> all-gather on the other") if pipe is not None else None),
                       "steps_per_allgather": gather_every if (pipelined and multi) else (1 if multi else None),
                       "steps_per_allgather_note": ((f"the bindings of {gather_every} consecutive batches travel in ONE RCCL all-gather" if gather_every > 1 else
                                                     "one RCCL all-gather of the bindings per batch (north_star's step); config.allgather_every_4 is the same loop with "
                                                     "four batches' bindings per collective") if (multi and pipelined) else None),
                       "allgather": (("torch.distributed.all_gather_into_tensor" if args.torch_gather else "ksched_allgather_bindings (C ABI, ncclAllGather on the pick's stream)") if multi else None),
                       "two_batches_in_flight": overlapped, "allgather_every_4": alt, "no_allgather": solo, "allgather_fallback": rig.comm_note,
                       "scaling_efficiency_vs_no_allgather": (value / (world * per_gpu_solo) if per_gpu_solo else None),
                       "configs3_strong": strong,
                       "parallelism": f"pod-row shards x{world}, node snapshot replicated, "
                       "allgather(int32 bindings)" if world > 1 else "single GPU",
                       "bound_fraction": bound_frac, "parity_check": parity,
                       "snapshot_refresh": ({"every_steps": args.refresh_every, "nodes_per_update": args.refresh_nodes,
                                             "updates_issued": refresh["calls"]} if args.refresh_every > 0 else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "traffic_committed": traffic_committed, "traffic_live": traffic_live,
                         "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
                         "kernel": f"mask kernel ({kernel_name}" + (", the sampled pick rides in it)" if pick_how.startswith("fused") else ")"),
                         "algorithmic_bytes_per_launch": alg,
                         "algorithmic_bytes_note": ("SURVEY.md 8d: P*b_pod + N*b_node + P*W*8 + P*4 with b_pod = 16 + 4*keys"
                                                    + (" + 8" if taint else "") + (f" + 4*{attempts_in_kernel} (the pod's draws: the sampled pick runs in this launch and writes the "
                                                       f"P*4 bindings); mask-only terms alone: {alg_mask_only}" if attempts_in_kernel else "; no pick in this launch: no draws, no bindings")),
                         "avg_kernel_us": avg_kernel_s * 1e6, "median_kernel_us": float(np.median(samples_us)) if launches else None,
                         "min_kernel_us": float(samples_us[0]) if launches else None, "max_kernel_us": float(samples_us[-1]) if launches else None,
                         "launches_timed": int(launches),
                         "timing": f"HIP events on every mask kernel dispatch of a post-pass of {launches} steps after the timed region "
                                   f"(clock ramp {ramp_steps} untimed steps >= {args.ramp_ms:.0f} ms before it); outputs rotate over {loop.R} mask buffer(s)",
                         "mask_kernel_evals_per_s": (hi - lo) * N / avg_kernel_s if avg_kernel_s > 0 else 0.0},
        }
        out["parity_check"] = parity
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(c, flag_names, p_cap=P_total)
        else:
            out["cpu_baseline"] = None
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    loop.close()
    rig.close()
    if rank == 0:
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)  # RCCL's version banner sits in the C stdio buffer until exit: push it out first
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)  # the ONE line, last
    if parity is not None and (parity.get("mismatches_all_ranks") or parity.get("mismatches") or 0) > 0:
        sys.stderr.write(f"bench.py: the self-check FAILED on rank {rank}: {parity}\n")
        sys.exit(1)
    # a checker that could not run leaves the number unverified: that is a failure too, unless explicitly allowed
    if parity is not None and not args.allow_unchecked and (parity.get("mismatches") is None or (parity.get("ranks_unchecked") or 0) > 0):
        sys.stderr.write(f"bench.py: the self-check could NOT RUN on rank {rank} (the line above is unverified; --allow-unchecked accepts that): {parity}\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
