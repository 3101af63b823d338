OUT=gpurun_out/r6w; mkdir -p $OUT
bash tools/box_fingerprint.sh > $OUT/box.txt 2>&1; grep -i "unique" $OUT/box.txt
for rep in 1 2 3; do for v in base pre8 pre16; do
  KSCHED_LIB=$PWD/build/variants/libksched_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --live-traffic off --no-others --repeats 1 2>/dev/null | tail -1 > $OUT/ab_${v}_$rep.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_${v}_$rep.json")); r=d["roofline"]
    print("$v rep $rep: step %.2f us  kernel %.2f us  frac %.3f  parity %s" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], (d.get("parity_check") or {}).get("mismatches")))
except Exception as e:
    print("$v: FAILED", e)
PY
done; done
KSCHED_LIB=$PWD/build/variants/libksched_hip_pre16.so python tools/trace_fused.py --workload C3 --pick --rotate 6 2>&1 | grep "d\[entry\|d\[staged\|d\[barrier\|d\[phase1"
KSCHED_LIB=$PWD/build/variants/libksched_hip_base.so python tools/trace_fused.py --workload C3 --pick --rotate 6 2>&1 | grep "d\[entry\|d\[staged\|d\[barrier\|d\[phase1"
