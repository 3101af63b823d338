#!/bin/bash
# Round-3 session K: does the L1 stall on hits to a pending line?  Microbenchmark, then the three kernels that put several requests to one
# line in flight (best-fit first stage: search blocks, candidate word pairs; sampled pick: record units, draws), A/B under KSCHED_OPT_DEBUG bits 25-26.
TAG=${1:-r3k}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
B25=$((1<<25)); B26=$((1<<26)); B=$((B25|B26))
stamp "microbenchmark"
timeout 120 ./tools/ubench_pending 2>&1 | tee $OUT/ubench_pending.txt
stamp "best-fit (bindings-only, C5 shard): shipped / searches one request per line / word pairs / both / both + one word per trip off"
timeout 600 python tools/bestfit_ab.py 0 $B25 $B26 $B 0 $B 2>&1 | tee $OUT/bestfit_ab.txt | tail -8
stamp "rocprofv3 split, both bits"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bf -o bf -- python $REPO/tools/bestfit_ab.py --no-oracle --steps 100 $B > $OUT/prof_bf.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $OUT/prof_bf -name "*.db" | head -1) 2>/dev/null | head -4
cd $REPO
stamp "sampled pick: C3 / C4s / C2, forms separate(0) waves(2) tile(3), debug 0 / bit25 / bit26 / both"
for wl in C3 C4s C2; do
 for fp in 0 2 3; do
  [ $wl = C2 ] && [ $fp = 3 ] && continue
  for dbg in 0 $B25 $B26 $B; do
   timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-others --no-strong-leg --fused-pick $fp --debug $dbg --steps 1500 2>$OUT/err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('%-4s fused-pick=%d debug=%-10s step %.2f us  kernel %.2f  pick=%s' % ('$wl', $fp, '$dbg', d['ms_per_step']*1e3, d['roofline'].get('avg_kernel_us') or -1, d.get('pick_launch')))
except Exception as e: print('$wl $fp $dbg FAILED', e)"
  done
 done
done
stamp "parity under both bits: pick tests"
KSCHED_DEBUG=$B timeout 900 python -m pytest tests/test_gpu_fused_pick.py tests/test_gpu_bestfit_deep.py -x -q 2>&1 | tail -2
stamp "done"
