cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.get_device_name(0))"
bash tools/gpu_ab.sh abA "s0p0 s1p0 s0p1 s1p1 s0p2 s1p2 s1p3" "C3 C4s"
bash tools/gpu_ab.sh abA5 "s0p0 s1p0 s1p1 s1p2" "C5s"
KSCHED_LIB=$PWD/build/variants/libksched_hip_s1p0.so timeout 200 python tools/trace_fused.py --workload C3 > gpurun_out/abA/trace_s1p0.txt 2>&1; head -16 gpurun_out/abA/trace_s1p0.txt
KSCHED_LIB=$PWD/build/variants/libksched_hip_s1p2.so timeout 200 python tools/trace_fused.py --workload C3 > gpurun_out/abA/trace_s1p2.txt 2>&1; head -16 gpurun_out/abA/trace_s1p2.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/abA/pytest_gpu.log 2>&1; tail -15 gpurun_out/abA/pytest_gpu.log
