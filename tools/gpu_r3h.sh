#!/bin/bash
# Round-3 session H: what the tile-test pick's parts cost (debug bits 27 / 28), C3 rotated.
TAG=${1:-r3h}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
line() {
  local label=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-others "$@" 2>$OUT/err_$label.log | tail -1 > $OUT/b_$label.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$label.json")); r=d["roofline"]; c=d["config"]
    print("%-22s step %6.2f us  kernel %6.2f (med %.2f) pick=%s" % ("$label", d["ms_per_step"]*1e3, r["avg_kernel_us"], r["median_kernel_us"] or 0, c.get("pick_launch")))
except Exception as e:
    print("$label: FAILED", e); print(open("$OUT/err_$label.log").read()[-1500:])
PY
}
line C3_tile --fused-pick 3
line C3_tile_notests --fused-pick 3 --debug 134217728
line C3_tile_noatomic --fused-pick 3 --debug 268435456
line C3_tile_neither --fused-pick 3 --debug 402653184
line C3_sep --fused-pick 0
line C4s_tile --workload C4s --fused-pick 3
line C4s_tile_notests --workload C4s --fused-pick 3 --debug 134217728
line C4s_tile_noatomic --workload C4s --fused-pick 3 --debug 268435456
line C4s_tile_neither --workload C4s --fused-pick 3 --debug 402653184
line C4s_sep --workload C4s --fused-pick 0
