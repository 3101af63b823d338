#!/bin/bash
# A/B timing of the fused-kernel build variants (tools/build_variants.sh) on the GPU box.
#   usage: bash tools/gpu_ab.sh <tag> "<variants>" "<workloads>"     e.g.  gpu_ab.sh ab1 "s0p0 s1p0 s1p2" "C3 C4s"
TAG=${1:-ab}; VARS=${2:-"s0p0 s1p0"}; WLS=${3:-"C3 C4s"}
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in $VARS; do for wl in $WLS; do
  KSCHED_LIB=$REPO/build/variants/libksched_hip_$v.so timeout 300 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --live-traffic off 2>&1 | tail -1 > $OUT/ab_${v}_${wl}.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_${v}_${wl}.json")); r=d["roofline"]
    print("$v $wl: step %.1f us  kernel %.2f us  frac %.3f  bound %.3f" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], d["config"]["bound_fraction"]))
except Exception as e:
    print("$v $wl: FAILED", e)
PY
done; done 2>&1 | tee $OUT/ab_summary.txt
