#!/bin/bash
# Round-3 session A: the riding pick (correctness first), then A/B timings of the variants behind debug bits.
#   usage: bash tools/gpu_r3a.sh <tag>
TAG=${1:-r3a}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
line() {  # line <label> <bench args...>
  local label=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-others "$@" 2>$OUT/err_$label.log | tail -1 > $OUT/b_$label.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$label.json")); r=d["roofline"]; c=d["config"]
    print("%-28s step %6.2f us  kernel %6.2f us (med %.2f)  frac %.3f  step_frac %s  pick=%s rot=%s bound=%.3f rep=%s" % ("$label", d["ms_per_step"]*1e3, r["avg_kernel_us"], r["median_kernel_us"] or 0, r["frac"], ("%.3f" % c["step_frac_of_hbm_peak"]) if c.get("step_frac_of_hbm_peak") else "-", c.get("pick_launch"), c.get("mask_rotation"), c["bound_fraction"], ["%.1f" % (x*1e3) for x in c.get("repeat_ms_per_step", [])]))
except Exception as e:
    print("$label: FAILED", e); print(open("$OUT/err_$label.log").read()[-1500:])
PY
}
stamp "new tests first: riding pick, fault injection"
timeout 600 python -m pytest tests/test_gpu_fused_pick.py -x -q > $OUT/pytest_fused_pick.log 2>&1; tail -5 $OUT/pytest_fused_pick.log
stamp "parity + golden + fullsize"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1; tail -5 $OUT/pytest_parity.log
stamp "C3: riding pick vs separate pick, rotation vs in place"
line C3_ride_rot
line C3_sep_rot --fused-pick 0
line C3_ride_inplace --no-rotate
line C3_sep_inplace --fused-pick 0 --no-rotate
stamp "C3: variants (pick waves last: 0x10000; setprio: 0x100000 / 0x200000)"
line C3_ride_last --debug 65536
line C3_prio --debug 1048576
line C3_prio_first --debug 2097152
line C3_prio_sep --debug 1048576 --fused-pick 0
stamp "C2: chunk divisors (bits 18-19: 1 = 16 rounds/block, 2 = 4, 3 = 1) ride vs separate"
line C2_ride --workload C2
line C2_sep --workload C2 --fused-pick 0
line C2_ride_d16 --workload C2 --debug 262144
line C2_ride_d1 --workload C2 --debug 786432
line C2_sep_d4 --workload C2 --fused-pick 0 --debug 524288
stamp "C4s"
line C4s_ride --workload C4s
line C4s_sep --workload C4s --fused-pick 0
line C4s_prio --workload C4s --debug 1048576
stamp "default line with others (C4s, C5s in-process) + cpu baseline, driver-style steps"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.log 2>&1; tail -1 $OUT/bench_driver_style.log > $OUT/bench_driver_style.json
python - <<PY
import json
d=json.load(open("$OUT/bench_driver_style.json")); c=d["config"]
print("driver-style: step %.2f us value %.3e frac %.3f step_frac %.3f in_place=%s" % (d["ms_per_step"]*1e3, d["value"], d["roofline"]["frac"], c["step_frac_of_hbm_peak"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in (c["in_place"] or {}).items() if k!="note"}))
for k,v in (c["other_workloads"] or {}).items():
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ("workload","pick_alone_note")})
PY
stamp "rocprofv3 kernel stats of the default command"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -o r -- python $REPO/bench.py --no-cpu-baseline --no-others > $OUT/prof_default.log 2>&1
cd $REPO
f=$(find $OUT/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prof_default_kernel_stats.csv && head -8 $f
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -size +2M -delete
stamp "done"
