#!/bin/bash
# quick check after a kernel change: the pick / parity tests, then bench lines of the four workloads (+ rocprof stats of C3)
OUT=$PWD/gpurun_out/${1:-q}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "${2:-pick or sampled or select or golden or c1_ or c2_ or c3_ or explain or fullsize}" 2>&1 | tail -3
for wl in C2 C3 C3h C4s C5s; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_$wl.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$wl.json")); r=d["roofline"]
    print("$wl: %.3e evals/s  step %.1f us  mask kernel %.2f us (median %.2f)  frac %.3f" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["median_kernel_us"], r["frac"]))
except Exception as e:
    print("$wl: FAILED", e)
PY
done
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python $REPO/bench.py --no-cpu-baseline > $OUT/prof.log 2>&1; cd $REPO
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4,6-7 $f | sed 's/(.*)"/"/' | head -4
find $OUT -name "*kernel_trace.csv" -delete
