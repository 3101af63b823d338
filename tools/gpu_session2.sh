#!/bin/bash
# usage: bash tools/gpu_session2.sh <tag> "<workloads>" "<kernels>"   (tests + bench per kernel + rocprof csv stats)
TAG=${1:-s}; WLS=${2:-"C3 C4s C5s C2"}; KERNELS=${3:-"indexed"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench"
for k in $KERNELS; do for wl in $WLS; do
  timeout 600 python bench.py --workload $wl --kernel $k --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_${wl}_${k}.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${wl}_${k}.json")); r=d["roofline"]
    print("$wl $k: %.3e evals/s  step %.1f us  kernel %.1f us  %.0f GB/s  frac %.3f" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["achieved"], r["frac"]))
except Exception as e:
    print("$wl $k: FAILED", e); print(open("$OUT/bench_${wl}_${k}.json").read()[-2000:])
PY
done; done
if [ -n "$PROF" ]; then
  echo "== rocprofv3 kernel stats"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o c3 -- python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
  cd $REPO
  f=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -8
fi
true
