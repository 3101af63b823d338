#!/bin/bash
# A/B of the HIP event flags used for the per-dispatch kernel timing (bench.py post-pass) against rocprofv3's kernel trace.
OUT=$PWD/gpurun_out/${1:-evflags}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for f in 0x0 0x20000000 0x40000000; do
  KSCHED_TIMING_EVENT_FLAGS=$f timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_$f.json
  python - <<PY
import json
r=json.load(open("$OUT/bench_$f.json"))["roofline"]
print("flags $f: mean %.2f median %.2f min %.2f max %.2f us" % (r["avg_kernel_us"], r["median_kernel_us"], r["min_kernel_us"], r["max_kernel_us"]))
PY
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python $REPO/bench.py --no-cpu-baseline > $OUT/prof.log 2>&1
cd $REPO
head -3 $OUT/prof/r_kernel_stats.csv | cut -c1-60,200-
grep -o '"avg_kernel_us": [0-9.]*\|"median_kernel_us": [0-9.]*' $OUT/prof.log
find $OUT -name "*kernel_trace.csv" -delete
