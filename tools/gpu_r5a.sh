#!/bin/bash
# round 5, session A: (1) new pipe/grid tests, (2) batches in flight on shares of the chip, (3) the fixed cost of a short region, (4) 64-byte store pattern
OUT=$PWD/gpurun_out/r5a; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_fused_pick.py -m gpu -x -q -k "pipe or grid_cus" 2>&1 | tail -5 | tee $OUT/pytest_pipe.log
echo "== inflight sweep C3"; timeout 400 python tools/inflight_sweep.py C3 2000 20 2>&1 | tee $OUT/inflight_C3.txt
echo "== inflight sweep C3 (pick forced to tile tests)"; SWEEP="4:64,6:48,8:32" timeout 300 python tools/inflight_sweep.py C3 2000 20 3 2>&1 | tee $OUT/inflight_C3_tilepick.txt
echo "== inflight sweep C4s"; SWEEP="1:0,2:0,2:128,3:88,4:64,4:96" timeout 400 python tools/inflight_sweep.py C4s 1000 20 2>&1 | tee $OUT/inflight_C4s.txt
echo "== fixed cost"; timeout 900 python tools/fixed_cost.py C3 2>&1 | tee $OUT/fixed_cost_C3.txt
echo "== ubench3"; timeout 200 tools/ubench3 2>&1 | tee $OUT/ubench3.txt
