cd $GRAFT_REPO_ROOT
./tools/ubench3 | tee gpurun_out/ubench3.txt
bash tools/gpu_ab.sh abC "p0 p2 p2t512 p2t768" "C3 C4s C5s"
KSCHED_LIB=$PWD/build/variants/libksched_hip_p2.so timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/abC/pytest_gpu_p2.log 2>&1; tail -5 gpurun_out/abC/pytest_gpu_p2.log
KSCHED_LIB=$PWD/build/variants/libksched_hip_p2t512.so timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/abC/pytest_gpu_p2t512.log 2>&1; tail -5 gpurun_out/abC/pytest_gpu_p2t512.log
