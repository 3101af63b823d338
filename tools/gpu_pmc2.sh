#!/bin/bash
# PMC counters of the fused kernel for an arbitrary command.  usage: bash tools/gpu_pmc2.sh <tag> "<counters>" -- <cmd...>
TAG=$1; CTRS=$2; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    if "k_eval" not in k: continue
    print(k, {c: round(sum(v)/len(v),1) for c,v in d.items()}, "n=%d"%len(next(iter(d.values()))))
PY
