#!/bin/bash
# r02c: snapshot build on the device: costs, rocprof of the build kernels, default bench
OUT=$PWD/gpurun_out/r02c; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_index_build.py -x -q 2>&1 | tail -2
timeout 300 python tools/host_costs.py > $OUT/host_costs.txt 2>&1; cat $OUT/host_costs.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_build -o r -- python $REPO/tools/host_costs.py > $OUT/prof_build.log 2>&1; cd $REPO
cut -d, -f1-4,6-7 $OUT/prof_build/r_kernel_stats.csv | sed 's/(.*)"/"/' | head -20
find $OUT -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_C3.json; python -c "
import json; d=json.load(open('$OUT/bench_C3.json')); print('C3: %.1f us/step kernel %.2f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['roofline']['frac']))"
