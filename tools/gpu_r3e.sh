#!/bin/bash
# Round-3 session E: best-fit (hint-routed second stage, LDS search levels, no memset); staging before the operand loads (bit 26).
TAG=${1:-r3e}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
line() {
  local label=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-others "$@" 2>$OUT/err_$label.log | tail -1 > $OUT/b_$label.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$label.json")); r=d["roofline"]; c=d["config"]
    print("%-22s step %6.2f us  kernel %6.2f (med %.2f) frac %.3f step_frac %s pick=%s rot=%s" % ("$label", d["ms_per_step"]*1e3, r["avg_kernel_us"], r["median_kernel_us"] or 0, r["frac"], ("%.3f" % c["step_frac_of_hbm_peak"]) if c.get("step_frac_of_hbm_peak") else "-", c.get("pick_launch"), c.get("mask_rotation")))
except Exception as e:
    print("$label: FAILED", e); print(open("$OUT/err_$label.log").read()[-1500:])
PY
}
stamp "tests: best fit (coarse, parity best-fit tests, list keys, fullsize C5s)"
timeout 1200 python -m pytest tests/test_gpu_bestfit_coarse.py tests/test_gpu_list_keys.py tests/test_gpu_fullsize.py "tests/test_gpu_parity.py" -x -q -m gpu -k "bestfit or coarse or list or fullsize or c5 or ragged or scratch" > $OUT/pytest_bf.log 2>&1; tail -3 $OUT/pytest_bf.log
stamp "C5s bindings-only: default / no summaries (bit 11) ; hand-over after 4 / 8 / 12 words"
for dbg in 0 2048 16384 49152; do
  timeout 200 python bench.py --workload C5s --no-cpu-baseline --no-others --no-mask --debug $dbg --steps 300 2>$OUT/err_c5_$dbg.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('C5s debug=%-9s bindings-only step %.1f us' % ('$dbg', d['ms_per_step']*1e3))
except Exception as e: print('C5s debug=$dbg FAILED', e)"
done
line C5s_full --workload C5s --steps 200
stamp "rocprofv3, C5s bindings-only (default)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5s_pick -o r -- python $REPO/bench.py --workload C5s --no-cpu-baseline --no-others --no-mask --steps 300 > $OUT/prof_c5s_pick.log 2>&1
cd $REPO
f=$(find $OUT/prof_c5s_pick -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prof_c5s_pick_kernel_stats.csv && head -5 $f | cut -c1-160
stamp "staging before the operand loads (bit 26 = 67108864): C3 ride / sep, C4s, C5s mask"
line C3_ride
line C3_ride_early --debug 67108864
line C3_sep --fused-pick 0
line C3_sep_early --fused-pick 0 --debug 67108864
line C3_sep_inplace --fused-pick 0 --no-rotate
line C3_sep_inplace_early --fused-pick 0 --no-rotate --debug 67108864
line C4s_ride --workload C4s
line C4s_ride_early --workload C4s --debug 67108864
line C5s_early --workload C5s --steps 200 --debug 67108864
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -size +2M -delete
stamp "done"
