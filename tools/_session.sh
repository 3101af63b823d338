OUT=gpurun_out/r6m; mkdir -p $OUT
bash tools/box_fingerprint.sh > $OUT/box.txt 2>&1; grep -i "unique" $OUT/box.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -3
timeout 300 python tools/fuzz_parity.py 120 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
timeout 600 python tools/alloc_probe.py C5s survey k=4 passes=3 hows=1,4,11 frag=0 > $OUT/survey_C5s.txt 2>&1; grep "^==\|failed\|Error\|library" $OUT/survey_C5s.txt | cut -c1-250
timeout 600 python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json")); r=d["roofline"]; g=d["config"]
print("default: %.3e evals/s step %.2f us kernel %.2f us frac %.3f traffic %s parity %s" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], r.get("traffic"), (d.get("parity_check") or {}).get("mismatches")))
for k, v in (g.get("other_workloads") or {}).items():
    print("   ", k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("workload", "pick_alone_note")})
PY
