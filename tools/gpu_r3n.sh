#!/bin/bash
# Round-3 session N: best-fit hand-over through per-wave masks (no counter); second-stage block size; hand-over point re-swept; traces.
TAG=${1:-r3n}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
B128=$((1<<21)); B64=$((2<<21))
stamp "best-fit (bindings-only, C5 shard): shipped(256) / 128 / 64 / hand-over after 4 6 10 12 16 words(256) / 12+128 / traced"
KSCHED_BF_TRACE_FILE=$OUT/trace.bin timeout 600 python tools/bestfit_ab.py 0 $B128 $B64 0x4000 0x6000 0xa000 0xc000 0xf000 $((0xc000|B128)) 0 0x100000 2>&1 | tee $OUT/bestfit_ab.txt | grep -v amdgpu.ids
stamp "parity: pick tests + list keys"
timeout 900 python -m pytest tests/test_gpu_fused_pick.py tests/test_gpu_bestfit_deep.py tests/test_gpu_parity.py tests/test_gpu_list_keys.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
stamp "fuzz 60 s"
timeout 200 python tools/fuzz_parity.py 60 11 2>&1 | tail -2
stamp "done"
