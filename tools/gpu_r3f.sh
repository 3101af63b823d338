#!/bin/bash
# Round-3 session F: full GPU suite; best-fit first stage two words per trip (A/B bit 11), hand-over points.
TAG=${1:-r3f}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
stamp "full GPU suite"
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest_gpu.log | tail -3
stamp "C5s bindings-only: two words per trip (default) / one (bit 11); hand-over after 6 / 8 / 10 / 12 / 14 words"
for dbg in 0 2048 24576 26624 40960 49152 57344; do
  timeout 200 python bench.py --workload C5s --no-cpu-baseline --no-others --no-mask --debug $dbg --steps 300 2>$OUT/err_c5_$dbg.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('C5s debug=%-9s bindings-only step %.1f us' % ('$dbg', d['ms_per_step']*1e3))
except Exception as e: print('C5s debug=$dbg FAILED', e)"
done
stamp "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
stamp "done"
