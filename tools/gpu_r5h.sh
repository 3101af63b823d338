#!/bin/bash
# round 5, session H: sc1 nt (p4) against sc0 sc1 nt (p5) against sc1 (p2): more repetitions, the driver's form, the in-place and two-stream legs
OUT=$PWD/gpurun_out/r5h; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do
for v in p2 p4 p5; do
  for wl in C3 C4s C5s; do
  st=2000; [ $wl = C5s ] && st=300
  KSCHED_LIB=$PWD/build/variants/libksched_hip_$v.so timeout 300 python bench.py --workload $wl --steps $st --no-cpu-baseline --no-others --repeats 3 2>/dev/null | tail -1 > $OUT/ab_${v}_${wl}_$rep.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_${v}_${wl}_$rep.json")); r=d["roofline"]; c=d["config"]
    print("$v $wl rep $rep: step %.2f us  kernel %.2f us  frac %.3f  repeats %s parity %s in_place %s two %s" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], [round(x*1e3,2) for x in c["repeat_ms_per_step"]], d["parity_check"]["mismatches"], c["in_place"] and round(c["in_place"]["ms_per_step"]*1e3,2), c["two_batches_in_flight"] and round(c["two_batches_in_flight"]["ms_per_step"]*1e3,2)))
except Exception as e:
    print("$v $wl: FAILED", e)
PY
  done
  KSCHED_LIB=$PWD/build/variants/libksched_hip_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/drv_${v}_$rep.json
  python - <<PY
import json
d=json.load(open("$OUT/drv_${v}_$rep.json")); r=d["roofline"]; c=d["config"]
print("$v driver form rep $rep: step %.2f us kernel %.2f us repeats %s in_place %.2f two %.2f C4s %.1f C5s %.1f C3x4 %.1f" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], [round(x*1e3,2) for x in c["repeat_ms_per_step"]], c["in_place"]["ms_per_step"]*1e3, c["two_batches_in_flight"]["ms_per_step"]*1e3, c["other_workloads"]["C4s"]["ms_per_step"]*1e3, c["other_workloads"]["C5s"]["ms_per_step"]*1e3, c["other_workloads"]["C3x4"]["ms_per_step"]*1e3))
PY
done; done 2>&1 | tee $OUT/ab_summary.txt
