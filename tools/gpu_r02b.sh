#!/bin/bash
# r02b: RCCL behind the C ABI (one-rank tests) + the N > 1 bench path in a one-rank group (KSCHED_BENCH_FORCE_DIST=1)
OUT=$PWD/gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "allgather or rccl" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for wl in C3 C4s; do
  for mode in "" "--torch-gather" "--one-stream"; do
    KSCHED_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload $wl --no-cpu-baseline $mode 2>&1 | tail -1 > $OUT/dist_${wl}_${mode#--}.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/dist_${wl}_${mode#--}.json")); c=d["config"]
    print("$wl $mode: gather/step: %.1f us/step  %.3e evals/s | every 4: %s | kernel %.1f us" % (d["ms_per_step"]*1e3, d["value"], (c["allgather_every_4"] or {}).get("ms_per_step"), d["roofline"]["avg_kernel_us"]))
except Exception as e:
    print("$wl $mode FAILED", e); print(open("$OUT/dist_${wl}_${mode#--}.json").read()[-600:])
PY
  done
done
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $OUT/single_C3.json; python -c "
import json; d=json.load(open('$OUT/single_C3.json')); print('single C3: %.1f us/step kernel %.2f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['roofline']['frac']))"
