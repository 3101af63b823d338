OUT=gpurun_out/r6t; mkdir -p $OUT
bash tools/box_fingerprint.sh > $OUT/box.txt 2>&1; grep -i "unique" $OUT/box.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $OUT/pytest_gpu.log | tail -5; tail -30 $OUT/pytest_gpu.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -15
