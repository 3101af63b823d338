#!/bin/bash
# ablation of the indexed kernel: bench.py --debug bits (1 no stores, 2 no search, 4 no sel loads, 8 no staging, 16 no main loop)
WLS=${1:-"C3 C4s"}; BITS=${2:-"0 1 2 4 8 16 6 22 30 31"}
for wl in $WLS; do for d in $BITS; do
  timeout 300 python bench.py --workload $wl --kernel indexed --steps 30 --warmup 3 --no-cpu-baseline --debug $d 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl debug=%2d: kernel %.1f us  step %.1f us' % ($d, r['avg_kernel_us'], d['ms_per_step']*1e3))"
done; done
