cd $GRAFT_REPO_ROOT
KSCHED_LIB=$PWD/build/variants/libksched_hip_lad2.so timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
bash tools/gpu_ab.sh abL "lad0 lad2 lad0 lad2" "C3 C4s"
KSCHED_LIB=$PWD/build/variants/libksched_hip_lad2.so timeout 200 python tools/trace_fused.py --workload C3 2>&1 | grep -v amdgpu | head -14
