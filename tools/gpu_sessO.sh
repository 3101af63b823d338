cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for mode in "" "--one-stream" "--depth 1"; do for wl in C3 C4s; do timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline $mode 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$mode] $wl: value %.3e step %.1f us kernel %.2f us frac %.3f two_stream %s' % (d['value'], d['ms_per_step']*1e3, r['avg_kernel_us'], r['frac'], d['config']['two_stream']))"; done; done
for mode in "" "--one-stream" "--gather-every 8"; do KSCHED_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 96 --warmup 8 --no-cpu-baseline $mode 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[force-dist $mode]: value %.3e step %.1f us kernel %.2f us in_flight %s per_gather %s two_stream %s' % (d['value'], d['ms_per_step']*1e3, r['avg_kernel_us'], d['config']['steps_in_flight'], d['config']['steps_per_allgather'], d['config']['two_stream']))"; done
