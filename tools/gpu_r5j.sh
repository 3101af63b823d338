#!/bin/bash
# round 5, session J: the best-fit pick beside the mask kernel (C5 shard), A/B; its test
OUT=$PWD/gpurun_out/r5j; mkdir -p $OUT; export TMPDIR=/tmp
echo "== test"; timeout 600 python -m pytest tests/test_gpu_fused_pick.py -m gpu -x -q -k "beside" 2>&1 | tail -4
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5j/c5s_beside.txt
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L, synth
for name, steps in (("C5s", 300), ("C5hs", 200)):
    for beside in (0, 1, 0, 1):
        r = bench.SingleRig(torch, L, synth, Evaluator, torch.device("cuda:0"), name)
        r.ev.set_option(L.OPT_PICK_BESIDE_MASK, beside)
        m = r.measure(steps=steps, samples=16)
        print(f"{name} beside={beside}: step {m['ms_per_step']*1e3:7.2f} us  mask kernel {m['mask_kernel_us']:7.2f} us  pick alone {m['pick_alone_us_per_step']:6.2f} us  pick={r.ev.last_pick}", flush=True)
        r.close(); del r; torch.cuda.empty_cache()
PY
