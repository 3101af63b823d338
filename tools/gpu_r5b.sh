#!/bin/bash
# round 5, session B: does hipExtAnyOrderLaunch work on gfx950, and what does it buy the one-stream loop?
OUT=$PWD/gpurun_out/r5b; mkdir -p $OUT; export TMPDIR=/tmp
echo "== ubench_anyorder"; timeout 120 tools/ubench_anyorder 2>&1 | tee $OUT/ubench_anyorder.txt
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_fused_pick.py -m gpu -x -q -k "pipe or grid_cus" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/pytest_pipe.log
echo "== sweep C3"; SWEEP="1:0,1:0:2,1:0:4,1:0:8,1:0:16,2:0,2:0:8,1:0" timeout 400 python tools/inflight_sweep.py C3 2000 20 2>&1 | tee $OUT/anyorder_C3.txt
echo "== sweep C4s"; SWEEP="1:0,1:0:4,1:0:8,2:0" timeout 400 python tools/inflight_sweep.py C4s 1000 20 2>&1 | tee $OUT/anyorder_C4s.txt
echo "== sweep C2"; SWEEP="1:0,1:0:8,2:0" timeout 400 python tools/inflight_sweep.py C2 4000 20 2>&1 | tee $OUT/anyorder_C2.txt
