cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for wl in C3 C4s C2 C5s; do timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl: value %.3e step %.1f us kernel %.2f us frac %.3f bound %.3f' % (d['value'], d['ms_per_step']*1e3, r['avg_kernel_us'], r['frac'], d['config']['bound_fraction']))"; done
