#!/bin/bash
# Round-3 session C: best-fit second stage over the row summaries; host-mirror fault paths; full GPU suite.
TAG=${1:-r3c}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
stamp "new tests: coarse best fit, host mirror (gpu)"
timeout 900 python -m pytest tests/test_gpu_bestfit_coarse.py tests/test_host_mirror.py -x -q -m gpu > $OUT/pytest_new.log 2>&1; tail -6 $OUT/pytest_new.log
stamp "C5s bindings-only: second stage coarse (default) vs full rows (bit 11 = 2048), hand-over after 1 / 2 / 4 / 8 words (bits 12-15)"
for dbg in 0 2048 4096 6144 8192 10240 16384 18432 32768; do
  timeout 200 python bench.py --workload C5s --no-cpu-baseline --no-others --no-mask --debug $dbg --steps 300 2>$OUT/err_c5_$dbg.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('C5s debug=%-6s bindings-only step %.1f us  bound=%.3f' % ('$dbg', d['ms_per_step']*1e3, d['config']['bound_fraction']))
except Exception as e: print('C5s debug=$dbg FAILED', e)"
done
stamp "C5s full step, C5hs (list key) bindings-only"
timeout 300 python bench.py --workload C5s --no-cpu-baseline --no-others --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C5s full step %.1f us  mask kernel %.1f us frac %.3f step_frac %.3f' % (d['ms_per_step']*1e3, r['avg_kernel_us'], r['frac'], d['config']['step_frac_of_hbm_peak']))"
timeout 300 python bench.py --workload C5hs --no-cpu-baseline --no-others --no-mask --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C5hs bindings-only step %.1f us' % (d['ms_per_step']*1e3))"
stamp "rocprofv3 kernel stats, C5s full step"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5s -o r -- python $REPO/bench.py --workload C5s --no-cpu-baseline --no-others --steps 200 > $OUT/prof_c5s.log 2>&1
cd $REPO
f=$(find $OUT/prof_c5s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prof_c5s_kernel_stats.csv && head -8 $f | cut -c1-220
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -size +2M -delete
stamp "full GPU suite"
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest_gpu.log | tail -3
stamp "done"
