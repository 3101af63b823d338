cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for mode in "" "--time-every 1" "--depth 1"; do for wl in C3 C4s C5s C2; do timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline $mode 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$mode] $wl: value %.3e step %.1f us kernel %.2f us (%d timed) frac %.3f' % (d['value'], d['ms_per_step']*1e3, r['avg_kernel_us'], r['launches_timed'], r['frac']))"; done; done
