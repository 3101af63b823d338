#!/bin/bash
# Round-3 session L: best-fit pick with aligned word pairs (shipped), per-wave time stamps of its two stages, parity tests of the picks.
TAG=${1:-r3l}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
stamp "best-fit (bindings-only, C5 shard): shipped, traced, shipped"
timeout 600 python tools/bestfit_ab.py 0 0x100000 0 2>&1 | tee $OUT/bestfit_ab.txt | grep -v amdgpu.ids
stamp "parity: pick tests"
timeout 900 python -m pytest tests/test_gpu_fused_pick.py tests/test_gpu_bestfit_deep.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
stamp "done"
