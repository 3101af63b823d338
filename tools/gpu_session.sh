#!/bin/bash
# One gpurun session: smoke, GPU tests, micro-benchmarks, bench lines, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh <tag>
TAG=${1:-s}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
echo "== rocminfo" ; rocminfo | grep -E "Marketing Name|gfx" | head -4
nproc
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== ubench"
timeout 120 ./tools/ubench > $OUT/ubench.txt 2>&1 ; tail -60 $OUT/ubench.txt
echo "== bench"
for wl in C3 C2 C4s C5s; do
  timeout 600 python bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_${wl}.json
done
timeout 900 python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | tee $OUT/bench_default.json
echo "== rocprofv3 kernel stats (default bench command)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o c3 -- python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
cd $REPO
find $OUT/prof_stats -name "*stats*" | head; f=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
