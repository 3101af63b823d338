#!/bin/bash
# Round-3 session B: barrier-first pick + dynamic rounds; alternate-stream leg; phase trace with the riding pick.
TAG=${1:-r3b}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
line() {  # line <label> <bench args...>
  local label=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-others "$@" 2>$OUT/err_$label.log | tail -1 > $OUT/b_$label.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$label.json")); r=d["roofline"]; c=d["config"]
    ov=c.get("two_batches_in_flight") or {}
    print("%-24s step %6.2f us  kernel %6.2f (med %.2f) frac %.3f step_frac %s pick=%s rot=%s rep=%s %s" % ("$label", d["ms_per_step"]*1e3, r["avg_kernel_us"], r["median_kernel_us"] or 0, r["frac"], ("%.3f" % c["step_frac_of_hbm_peak"]) if c.get("step_frac_of_hbm_peak") else "-", c.get("pick_launch"), c.get("mask_rotation"), ["%.1f" % (x*1e3) for x in c.get("repeat_ms_per_step", [])], ("| 2 streams: %.2f us (%s, eq=%s)" % (ov["ms_per_step"]*1e3, ov.get("pick_launch"), ov.get("bindings_equal_sequential"))) if "ms_per_step" in ov else (ov.get("error","") if ov else "")))
except Exception as e:
    print("$label: FAILED", e); print(open("$OUT/err_$label.log").read()[-1500:])
PY
}
stamp "tests: riding pick, parity, golden, fullsize"
timeout 900 python -m pytest tests/test_gpu_fused_pick.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest.log | tail -3
stamp "C3 (rotated): ride (barrier first, dynamic rounds) / old order / dynamic rounds + separate pick / separate pick"
line C3_ride --overlap-leg
line C3_ride_oldorder --debug 8388608
line C3_dyn_sep --debug 4194304
line C3_sep --fused-pick 0 --overlap-leg
stamp "C3 in place"
line C3_ride_inplace --no-rotate
line C3_dyn_sep_inplace --no-rotate --debug 4194304
line C3_sep_inplace --no-rotate --fused-pick 0
stamp "C4s, C2"
line C4s_ride --workload C4s --overlap-leg
line C4s_dyn_sep --workload C4s --debug 4194304
line C4s_sep --workload C4s --fused-pick 0
line C2_ride --workload C2 --overlap-leg
line C2_sep --workload C2 --fused-pick 0
stamp "phase trace, C3, with and without the riding pick"
timeout 200 python tools/trace_fused.py --workload C3 --pick > $OUT/trace_C3_pick.txt 2>&1; head -16 $OUT/trace_C3_pick.txt
timeout 200 python tools/trace_fused.py --workload C3 > $OUT/trace_C3_nopick.txt 2>&1; head -15 $OUT/trace_C3_nopick.txt
timeout 200 python tools/trace_fused.py --workload C3 --pick --debug 8388608 > $OUT/trace_C3_pick_oldorder.txt 2>&1; head -16 $OUT/trace_C3_pick_oldorder.txt
stamp "driver-style default line"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.log 2>&1; tail -1 $OUT/bench_driver_style.log > $OUT/bench_driver_style.json
python - <<PY
import json
d=json.load(open("$OUT/bench_driver_style.json")); c=d["config"]
print("driver-style: step %.2f us value %.3e frac %.3f step_frac %.3f" % (d["ms_per_step"]*1e3, d["value"], d["roofline"]["frac"], c["step_frac_of_hbm_peak"]))
for k in ("in_place","two_batches_in_flight"):
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in (c[k] or {}).items() if kk!="note"})
for k,v in (c["other_workloads"] or {}).items():
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ("workload","pick_alone_note")})
PY
stamp "done"
