#!/bin/bash
# quick loop: GPU parity tests (optional) + bench lines for given workloads/kernels
# usage: bash tools/gpu_quick.sh <tag> "<workloads>" "<kernels>" [notest]
TAG=${1:-q}; WLS=${2:-"C3 C4s C5s C2"}; KERNELS=${3:-"fused"}; NOTEST=$4
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -z "$NOTEST" ]; then echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8; fi
for k in $KERNELS; do for wl in $WLS; do
  timeout 600 python bench.py --workload $wl --kernel $k --steps 30 --warmup 3 --no-cpu-baseline --live-traffic off $BENCH_EXTRA 2>&1 | tail -1 > $OUT/bench_${wl}_${k}.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${wl}_${k}.json")); r=d["roofline"]
    print("$wl $k: %.3e evals/s  step %.1f us  kernel %.1f us  %.0f GB/s  frac %.3f" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["achieved"], r["frac"]))
except Exception as e:
    print("$wl $k: FAILED", e); print(open("$OUT/bench_${wl}_${k}.json").read()[-1500:])
PY
done; done
