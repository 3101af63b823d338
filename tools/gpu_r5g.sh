#!/bin/bash
# round 5, session G: store policy A/B with rotated outputs AND inputs (round 1 chose sc1 with one buffer rewritten in place)
OUT=$PWD/gpurun_out/r5g; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for v in p2 p3 p0 p1 p4 p5; do for wl in C3 C4s C5s; do
  st=2000; [ $wl = C5s ] && st=300
  KSCHED_LIB=$PWD/build/variants/libksched_hip_$v.so timeout 300 python bench.py --workload $wl --steps $st --no-cpu-baseline --no-others --repeats 2 2>/dev/null | tail -1 > $OUT/ab_${v}_${wl}_$rep.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_${v}_${wl}_$rep.json")); r=d["roofline"]
    print("$v $wl rep $rep: step %.2f us  kernel %.2f us  frac %.3f  repeats %s parity %s" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], [round(x*1e3,2) for x in d["config"]["repeat_ms_per_step"]], d["parity_check"]["mismatches"]))
except Exception as e:
    print("$v $wl: FAILED", e)
PY
done; done; done 2>&1 | tee $OUT/ab_summary.txt
