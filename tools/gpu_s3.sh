#!/bin/bash
# session: tests + bench (indexed & direct) + rocprof kernel stats + ablation of the indexed kernel
TAG=${1:-s3}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
nproc; rocminfo | grep -E "Marketing Name" | head -2
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
PROF=1 bash tools/gpu_session2.sh $TAG "C3 C4s C5s C2" "indexed"
bash tools/gpu_session2.sh ${TAG}d "C3" "direct" | grep -v pytest | tail -3
echo "== ablation"
bash tools/gpu_ablate.sh "C3" "0 1 2 4 8 16"
