cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/sessI_pytest.log 2>&1; tail -5 gpurun_out/sessI_pytest.log | cut -c1-300
for wl in C3 C4s C2; do timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl: value %.3e step %.1f us kernel %.2f us frac %.3f bound %.3f' % (d['value'], d['ms_per_step']*1e3, r['avg_kernel_us'], r['frac'], d['config']['bound_fraction']))"; done
timeout 300 python bench.py --workload C3 --steps 50 --warmup 5 --no-cpu-baseline --no-mask 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3 bindings only: value %.3e step %.1f us' % (d['value'], d['ms_per_step']*1e3))"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sessI_prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; f=$(find $GRAFT_REPO_ROOT/gpurun_out/sessI_prof -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-200
