cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh abB "p0 p2 p1 p4 p5 p6 p2g13 p2g26 p2t512 p0 p2" "C3 C4s"
bash tools/gpu_ab.sh abB5 "p0 p2 p1 p4 p5 p2t512" "C5s"
for wl in C3 C4s; do KSCHED_LIB=$PWD/build/variants/libksched_hip_p2.so timeout 300 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --debug 32 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('p2 rr-map $wl: kernel %.2f us frac %.3f' % (r['avg_kernel_us'], r['frac']))"; done
KSCHED_LIB=$PWD/build/variants/libksched_hip_p2.so timeout 200 python tools/trace_fused.py --workload C3 > gpurun_out/abB/trace_p2.txt 2>&1; head -16 gpurun_out/abB/trace_p2.txt
KSCHED_LIB=$PWD/build/variants/libksched_hip_p2g26.so timeout 200 python tools/trace_fused.py --workload C3 > gpurun_out/abB/trace_p2g26.txt 2>&1; head -16 gpurun_out/abB/trace_p2g26.txt
KSCHED_LIB=$PWD/build/variants/libksched_hip_p2.so timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/abB/pytest_gpu_p2.log 2>&1; tail -5 gpurun_out/abB/pytest_gpu_p2.log
