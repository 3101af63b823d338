cd $GRAFT_REPO_ROOT
python tools/gpu_crash_probe.py 2>&1 | tail -12
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/sessD_pytest.log 2>&1; tail -5 gpurun_out/sessD_pytest.log
for wl in C3 C4s C5s C2; do timeout 300 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('new $wl: step %.1f us kernel %.2f us frac %.3f bound %.3f' % (d['ms_per_step']*1e3, r['avg_kernel_us'], r['frac'], d['config']['bound_fraction']))"; done
timeout 200 python tools/trace_fused.py --workload C3 > gpurun_out/sessD_trace_C3.txt 2>&1; head -16 gpurun_out/sessD_trace_C3.txt
