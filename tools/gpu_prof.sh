#!/bin/bash
# rocprofv3 kernel stats of one bench command.  usage: bash tools/gpu_prof.sh <tag> <workload> <kernel> [extra bench args]
TAG=${1:-p}; WL=${2:-C3}; K=${3:-fused}; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${WL}_${K} -o r -- python $REPO/bench.py --workload $WL --kernel $K --steps 50 --warmup 5 --no-cpu-baseline --live-traffic off "$@" > $OUT/prof_${WL}_${K}.log 2>&1
cd $REPO
tail -1 $OUT/prof_${WL}_${K}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$WL $K (profiled run): step %.1f us  event-kernel %.1f us' % (d['ms_per_step']*1e3, r['avg_kernel_us']))"
f=$(find $OUT/prof_${WL}_${K} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ksched' in r['Name']: print("  %-40s calls %s avg %.2f us (min %.2f max %.2f)" % (r['Name'].split('(')[0][:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
