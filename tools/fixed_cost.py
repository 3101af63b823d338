#!/usr/bin/env python
"""Where does the fixed cost of a short timed region go?  (VERDICT r4 "weak" 3 / "do this" 3: the driver's form, 20 steps between
synchronizes, reads (ms_per_step - kernel) x 20 = 71 us more than 20 kernels.)

For the sequential C3 loop (bench.py's N = 1 form, outputs rotated) this times regions of K = 1 .. 200 steps between torch.cuda.synchronize()
pairs, splitting each region into the host's launch loop and the closing synchronize, and fits  T(K) = fixed + K x per_step.  It does so in
a fresh process per runtime setting a latency-sensitive host could choose:
    default                         the runtime as it comes
    HSA_ENABLE_INTERRUPT=0          completion signals are polled, not interrupt-driven
    ROC_ACTIVE_WAIT_TIMEOUT=200     the HIP runtime spins that many us on a signal before it blocks
    hipDeviceScheduleSpin           hipSetDeviceFlags(hipDeviceScheduleSpin) before the context exists
    GPU_MAX_HW_QUEUES=1 / 2         fewer hardware queues
    HIP_FORCE_DEV_KERNARG=0 / 1     where the kernel-argument segment lives (host / device memory)
    AMD_DIRECT_DISPATCH=0           launches handed to the runtime's worker thread instead of written by the calling thread
(FC_ONLY=substr[,substr] runs the matching variants only)
usage: python tools/fixed_cost.py [workload=C3]            (parent: runs every variant)
"""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [
    ("default", {}, None),
    ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}, None),
    ("ROC_ACTIVE_WAIT_TIMEOUT=200", {"ROC_ACTIVE_WAIT_TIMEOUT": "200"}, None),
    ("hipDeviceScheduleSpin", {}, 1),
    ("hipDeviceScheduleYield", {}, 2),
    ("hipDeviceScheduleBlockingSync", {}, 4),
    ("GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"}, None),
    ("spin + ROC_ACTIVE_WAIT_TIMEOUT=200 + HSA_ENABLE_INTERRUPT=0", {"ROC_ACTIVE_WAIT_TIMEOUT": "200", "HSA_ENABLE_INTERRUPT": "0"}, 1),
    ("HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}, None),  # kernel arguments in host memory (the kernels' first scalar loads cross PCIe)
    ("HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}, None),  # ... in device memory
    ("AMD_DIRECT_DISPATCH=0", {"AMD_DIRECT_DISPATCH": "0"}, None),  # launches through the runtime's worker thread
]


def child():
    name = os.environ.get("FC_WORKLOAD", "C3")
    flags_dev = os.environ.get("FC_DEVICE_FLAGS")
    import numpy as np
    import torch
    if flags_dev:  # before the HIP context of device 0 exists
        path = None
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
        hip = ctypes.CDLL(path or "libamdhip64.so")
        rc = hip.hipSetDeviceFlags(ctypes.c_uint(int(flags_dev)))
        if rc:
            print(json.dumps({"error": f"hipSetDeviceFlags({flags_dev}) -> {rc}"}))
            return
    sys.path.insert(0, ROOT)
    import bench
    from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth
    cfg, P, N, flag_names, pick, desc = bench.WORKLOADS[name]
    c = synth.make_config(cfg, P=P, N=N)
    dev = torch.device("cuda:0")
    ev = Evaluator(0)
    ev.set_nodes(**c.node_columns())
    flags = sum(getattr(L, f) for f in flag_names) | (L.PICK_SAMPLED if pick == "sampled" else L.PICK_BESTFIT)
    taint = "TAINT" in flag_names
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d = (t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32) if c.n_keys else None, t(c.pod_tol, np.int64) if taint else None,
         t(c.samples, np.int32) if pick == "sampled" else None)
    m0 = ev.alloc_mask(P)
    R = bench.rotation_for(int(m0.stride(0)) * 8 * P, True)
    masks = [m0] + [ev.alloc_mask(P) for _ in range(R - 1)]
    out = torch.full((P,), -1, dtype=torch.int32, device=dev)
    run = ev.bind_eval_device(*d, flags, out_feasible=masks, out_bindings=[out])
    k = [0]

    def step():
        run(0, k[0] % R)
        k[0] += 1

    def region(K):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t2 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6

    for _ in range(3000):
        step()
    torch.cuda.synchronize()
    rows = {}
    for K in (1, 2, 5, 10, 20, 50, 100, 200):
        rs = sorted(region(K) for _ in range(25))
        med = rs[len(rs) // 2]
        rows[K] = {"total_us": med[0], "launch_loop_us": med[1], "sync_us": med[2], "min_total_us": rs[0][0]}
    # T(K) = fixed + K * per_step (least squares over the medians)
    Ks = np.array(sorted(rows), dtype=float)
    Ts = np.array([rows[int(K)]["total_us"] for K in Ks])
    A = np.vstack([np.ones_like(Ks), Ks]).T
    (fixed, per_step), *_ = np.linalg.lstsq(A, Ts, rcond=None)
    # an empty synchronize, and a synchronize after ONE tiny kernel (the wake-up alone)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        torch.cuda.synchronize()
    empty_sync = (time.perf_counter() - t0) / 200 * 1e6
    z = torch.zeros(64, device=dev)
    tiny = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z.add_(1.0)
        torch.cuda.synchronize()
        tiny.append((time.perf_counter() - t0) * 1e6)
    print(json.dumps({"fixed_us": float(fixed), "per_step_us": float(per_step), "driver_form_us_per_step": rows[20]["total_us"] / 20,
                      "empty_sync_us": empty_sync, "tiny_kernel_launch_to_sync_us": float(np.median(tiny)), "regions": rows}))


def main():
    if os.environ.get("FC_CHILD"):
        return child()
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    print(f"# tools/fixed_cost.py {name}: T(K steps between synchronizes) = fixed + K x per_step, sequential loop, outputs rotated; medians of 25 regions per K")
    only = os.environ.get("FC_ONLY")  # comma-separated substrings: run the matching variants only
    for label, env, devflags in VARIANTS:
        if only and not any(o in label for o in only.split(",")):
            continue
        e = dict(os.environ, FC_CHILD="1", FC_WORKLOAD=name, **env)
        if devflags is not None:
            e["FC_DEVICE_FLAGS"] = str(devflags)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode or not line:
            print(f"{label:62s} FAILED rc={r.returncode} {r.stderr[-300:]!r}")
            continue
        doc = json.loads(line[-1])
        if "error" in doc:
            print(f"{label:62s} {doc['error']}")
            continue
        g = doc["regions"]
        print(f"{label:62s} fixed {doc['fixed_us']:6.1f} us  per_step {doc['per_step_us']:6.2f} us  driver form (K=20) {doc['driver_form_us_per_step']:6.2f} us/step  "
              f"[K=20: launch loop {g['20']['launch_loop_us']:6.1f} + sync {g['20']['sync_us']:6.1f}]  K=1 total {g['1']['total_us']:5.1f}  "
              f"empty sync {doc['empty_sync_us']:4.1f}  tiny kernel {doc['tiny_kernel_launch_to_sync_us']:5.1f}", flush=True)


if __name__ == "__main__":
    main()
