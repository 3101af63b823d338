#!/bin/bash
# Round-3 session I: N > 1 path in a one-rank group: alternate (default) vs split pipe; the new alternate-mode tests.
TAG=${1:-r3i}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
stamp "tests"
timeout 900 python -m pytest tests/test_gpu_fused_pick.py tests/test_gpu_bench_contract.py -x -q -m gpu > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest.log | tail -3; grep -n "^E " $OUT/pytest.log | head
for wl in C3 C4s; do for mode in "" "--split-pipe" "--one-stream"; do
  KSCHED_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload $wl --no-cpu-baseline $mode 2>$OUT/err_${wl}_${mode#--}.log | tail -1 > $OUT/dist_${wl}_${mode#--}.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/dist_${wl}_${mode#--}.json")); c=d["config"]
    print("$wl %-14s %s | gather/step %.1f us/step %.3e evals/s | other gather cadence: %s | no gather: %s | mask kernel %.1f us | pick=%s" % ("$mode", (c.get("pipe_mode") or "-")[:9], d["ms_per_step"]*1e3, d["value"], round((c.get("allgather_every_step") or c.get("allgather_every_4") or {}).get("ms_per_step", 0)*1e3, 1), round((c.get("no_allgather") or {}).get("ms_per_step", 0)*1e3, 1), d["roofline"]["avg_kernel_us"], c.get("pick_launch")))
except Exception as e:
    print("$wl $mode FAILED", e); print(open("$OUT/err_${wl}_${mode#--}.log").read()[-1200:])
PY
done; done
KSCHED_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline 2>$OUT/dist_default.err | tail -1 > $OUT/dist_default.json
python - <<PY
import json
try:
    d=json.load(open("$OUT/dist_default.json")); c=d["config"]
    print("default N>1 path (one rank): %s | %s | gather/step %.1f us/step | no gather %s | eff vs no gather %s | strong leg %s" % (c["workload"][:24], (c.get("pipe_mode") or "-")[:9], d["ms_per_step"]*1e3, (c.get("no_allgather") or {}).get("ms_per_step"), c.get("scaling_efficiency_vs_no_allgather"), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (c.get("configs3_strong") or {}).items() if k not in ("workload", "no_allgather")}))
except Exception as e:
    print("default N>1 path FAILED", e); print(open("$OUT/dist_default.err").read()[-1500:])
PY
stamp "done"
