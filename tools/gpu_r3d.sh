#!/bin/bash
# Round-3 session D: eager draws in the riding pick; alternate-stream leg with the shipped kernel; adaptive best-fit second stage.
TAG=${1:-r3d}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
line() {
  local label=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-others "$@" 2>$OUT/err_$label.log | tail -1 > $OUT/b_$label.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$label.json")); r=d["roofline"]; c=d["config"]
    ov=c.get("two_batches_in_flight") or {}
    print("%-22s step %6.2f us  kernel %6.2f (med %.2f) frac %.3f step_frac %s pick=%s rot=%s %s" % ("$label", d["ms_per_step"]*1e3, r["avg_kernel_us"], r["median_kernel_us"] or 0, r["frac"], ("%.3f" % c["step_frac_of_hbm_peak"]) if c.get("step_frac_of_hbm_peak") else "-", c.get("pick_launch"), c.get("mask_rotation"), ("| 2 streams: %.2f us (%s, eq=%s)" % (ov["ms_per_step"]*1e3, ov.get("pick_launch"), ov.get("bindings_equal_sequential"))) if "ms_per_step" in ov else (ov.get("error","") if ov else "")))
except Exception as e:
    print("$label: FAILED", e); print(open("$OUT/err_$label.log").read()[-1500:])
PY
}
stamp "tests: coarse best fit (adaptive), riding pick"
timeout 900 python -m pytest tests/test_gpu_bestfit_coarse.py tests/test_gpu_fused_pick.py -x -q -m gpu > $OUT/pytest_new.log 2>&1; tail -3 $OUT/pytest_new.log
stamp "C3 riding pick: eager draws 1 (default) / 3 / 5 / 2; 2-stream alternate leg; pick waves last"
line C3_e1 --overlap-leg
line C3_e3 --debug 256
line C3_e5 --debug 512
line C3_e2 --debug 768
line C3_e5_last --debug 66048
line C3_sep --fused-pick 0 --overlap-leg
line C3_e1_inplace --no-rotate --overlap-leg
line C3_e5_inplace --no-rotate --debug 512
line C4s_e1 --workload C4s --overlap-leg
line C4s_e5 --workload C4s --debug 512
line C2_e1 --workload C2
line C2_e5 --workload C2 --debug 512
stamp "C5s bindings-only: summary scan adaptive (64 default / 16 / 256 / 1024 candidate bytes), full rows"
for dbg in 0 16777216 33554432 50331648 2048; do
  timeout 200 python bench.py --workload C5s --no-cpu-baseline --no-others --no-mask --debug $dbg --steps 300 2>$OUT/err_c5_$dbg.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('C5s debug=%-9s bindings-only step %.1f us' % ('$dbg', d['ms_per_step']*1e3))
except Exception as e: print('C5s debug=$dbg FAILED', e)"
done
stamp "rocprofv3, C5s bindings-only (default)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5s_pick -o r -- python $REPO/bench.py --workload C5s --no-cpu-baseline --no-others --no-mask --steps 300 > $OUT/prof_c5s_pick.log 2>&1
cd $REPO
f=$(find $OUT/prof_c5s_pick -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prof_c5s_pick_kernel_stats.csv && head -5 $f | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -size +2M -delete
stamp "done"
