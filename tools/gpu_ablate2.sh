#!/bin/bash
# ablation of the fused kernel.  usage: bash tools/gpu_ablate2.sh "<pods list>" "<debug bits list>" [workload]
PODS=${1:-"100000"}; BITS=${2:-"32 33 34 40 48"}; WL=${3:-C3}
for p in $PODS; do for d in $BITS; do
  timeout 300 python bench.py --workload $WL --kernel fused --steps 30 --warmup 3 --no-cpu-baseline --pods $p --debug $d 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$WL P=%6d debug=%2d: kernel %.1f us  step %.1f us' % ($p, $d, r['avg_kernel_us'], d['ms_per_step']*1e3))"
done; done
