#!/bin/bash
# Round-3 session O: best-fit hand-over through 128 sub-lists (one returning atomic per wave on a counter shared by ~15 waves), second stage one wave per block.
TAG=${1:-r3o}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
G1=$((1<<21)); G2=$((2<<21)); G8=$((3<<21))
stamp "best-fit (bindings-only, C5 shard): grid = capacity/4 (shipped) / full / half / eighth; hand-over after 4 6 10 12 words; traced"
KSCHED_BF_TRACE_FILE=$OUT/trace.bin timeout 600 python tools/bestfit_ab.py 0 $G1 $G2 $G8 0x4000 0x6000 0xa000 0xc000 0 0x100000 2>&1 | tee $OUT/bestfit_ab.txt | grep -v amdgpu.ids
stamp "parity: pick tests + list keys"
timeout 900 python -m pytest tests/test_gpu_fused_pick.py tests/test_gpu_bestfit_deep.py tests/test_gpu_parity.py tests/test_gpu_list_keys.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
stamp "fuzz 60 s"
timeout 200 python tools/fuzz_parity.py 60 13 2>&1 | tail -2
stamp "C5s full step"
timeout 300 python bench.py --workload C5s --no-cpu-baseline --no-others --no-strong-leg --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5s step %.1f us  mask kernel %.1f us' % (d['ms_per_step']*1e3, d['roofline'].get('avg_kernel_us') or -1))"
stamp "done"
