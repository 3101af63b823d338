#!/bin/bash
# Round-3 session G: the tile-test pick (PICK == 2): correctness, then C3 / C4s / C2 timings of the three forms.
TAG=${1:-r3g}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
line() {
  local label=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-others "$@" 2>$OUT/err_$label.log | tail -1 > $OUT/b_$label.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$label.json")); r=d["roofline"]; c=d["config"]
    ov=c.get("two_batches_in_flight") or {}
    print("%-22s step %6.2f us  kernel %6.2f (med %.2f) frac %.3f step_frac %s pick=%s rot=%s bound=%.3f %s" % ("$label", d["ms_per_step"]*1e3, r["avg_kernel_us"], r["median_kernel_us"] or 0, r["frac"], ("%.3f" % c["step_frac_of_hbm_peak"]) if c.get("step_frac_of_hbm_peak") else "-", c.get("pick_launch"), c.get("mask_rotation"), c["bound_fraction"], ("| 2 streams: %.2f us (%s, eq=%s)" % (ov["ms_per_step"]*1e3, ov.get("pick_launch"), ov.get("bindings_equal_sequential"))) if "ms_per_step" in ov else (ov.get("error","") if ov else "")))
except Exception as e:
    print("$label: FAILED", e); print(open("$OUT/err_$label.log").read()[-1500:])
PY
}
stamp "tests: riding pick (both forms), parity, fullsize"
timeout 1200 python -m pytest tests/test_gpu_fused_pick.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden.py -x -q -m gpu > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest.log | tail -3; grep -n "^E " $OUT/pytest.log | head -20
stamp "C3 rotated: tile / waves / separate"
line C3_tile --fused-pick 3 --overlap-leg
line C3_waves --fused-pick 2
line C3_sep --fused-pick 0
stamp "C3 in place"
line C3_tile_inplace --fused-pick 3 --no-rotate --overlap-leg
line C3_sep_inplace --fused-pick 0 --no-rotate
stamp "C4s, C2"
line C4s_tile --workload C4s --fused-pick 3 --overlap-leg
line C4s_waves --workload C4s --fused-pick 2
line C4s_sep --workload C4s --fused-pick 0
line C2_tile --workload C2 --fused-pick 3
line C2_waves --workload C2 --fused-pick 2
stamp "fuzz 40 s"
timeout 120 python tools/fuzz_parity.py 40 20260923 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
stamp "done"
