#!/bin/bash
# PMC passes for the mask kernel (separate runs, --pmc only + kernel-trace as the pool's gpurun requires)
TAG=${1:-pmc}; WL=${2:-C3}; K=${3:-fused}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|GRBM)_[A-Z0-9_]+\b" | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $REPO/bench.py --workload $WL --kernel $K --steps 10 --warmup 2 --no-cpu-baseline --live-traffic off $BENCH_EXTRA > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    if "k_eval" not in k and "k_pick" not in k: continue
    print(k, {c: round(sum(v)/len(v),1) for c,v in d.items()}, "n=%d"%len(next(iter(d.values()))))
PY
}
run pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
run pmc2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
run pmc3 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run pmc4 FETCH_SIZE
run pmc5 WRITE_SIZE
