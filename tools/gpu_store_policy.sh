#!/bin/bash
# A/B of the mask stores' cache policy inside the kernel (KSCHED_STORE_POLICY, tools/build_variants.sh sp0=-DKSCHED_STORE_POLICY=0 ... sp6=-DKSCHED_STORE_POLICY=6), alternating, per workload.
# usage: bash tools/gpu_store_policy.sh <tag> [workloads...]
TAG=${1:-sp}; shift; WLS=${@:-"C3 C4s C5s"}
O=gpurun_out/$TAG; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep -m1 Unique
for rep in 1 2; do for wl in $WLS; do for sp in 6 5 2 3 4 0 1; do
  KSCHED_LIB=$PWD/build/variants/libksched_hip_sp$sp.so timeout 300 python bench.py --workload $wl --steps 300 --warmup 10 --no-cpu-baseline --live-traffic off --no-others --repeats 0 2>/dev/null | tail -1 > $O/${wl}_sp${sp}_$rep.json
  python - <<PY
import json
try:
    d=json.load(open("$O/${wl}_sp${sp}_$rep.json")); r=d["roofline"]
    print("$wl policy $sp rep $rep: step %.2f us  kernel %.2f us  parity mismatches %s" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], d["parity_check"]["mismatches"]))
except Exception as e: print("$wl policy $sp FAILED", e)
PY
done; done; done
