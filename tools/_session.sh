OUT=gpurun_out/r7b; mkdir -p $OUT
bash tools/box_fingerprint.sh > $OUT/box.txt 2>&1; grep -i "unique" $OUT/box.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -3; grep -n "^FAILED\|^E  " $OUT/pytest_gpu.log | head -10
timeout 300 python tools/fuzz_parity.py 120 7 > $OUT/fuzz.txt 2>&1; tail -1 $OUT/fuzz.txt | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --live-traffic off 2>/dev/null | tail -1 > $OUT/bench.json
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); r=d["roofline"]; g=d["config"]
print("default: step %.2f us kernel %.2f us frac %.3f parity %s" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], d["parity_check"]["mismatches"]))
for k, v in (g.get("other_workloads") or {}).items():
    print("   ", k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("ms_per_step","mask_kernel_us","mask_kernel_frac","step_frac","pick_alone_us_per_step","pick_in_mask_launch")})
PY
