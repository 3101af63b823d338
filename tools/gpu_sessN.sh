cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "bestfit or fullsize or c5 or ragged" 2>&1 | grep -E "passed|failed|Error" | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sessN_prof -o r -- python $GRAFT_REPO_ROOT/bench.py --workload C5s --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1; f=$(find $GRAFT_REPO_ROOT/gpurun_out/sessN_prof -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ksched' in r['Name']: print("%-44s calls %s avg %.2f us (min %.2f max %.2f)" % (r['Name'].split('(')[0][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
