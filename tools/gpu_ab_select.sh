#!/bin/bash
# A/B of the sampled pick's eager-draw count (KSCHED_OPT_DEBUG bits 8-9): bindings-only bench (no mask kernel) at C3 and C2
export TMPDIR=/tmp
for wl in C3 C2; do for dbg in 0 256 512 768; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-mask --debug $dbg 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl debug=$dbg (eager %s): bindings-only step %.2f us' % ({0:2,256:3,512:5,768:1}[$dbg], d['ms_per_step']*1e3))"
done; done
