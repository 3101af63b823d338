#!/bin/bash
# Round-3 session J: best-fit second stage A/B (wider later rounds, smaller grid), rocprofv3 split + counters of the two best-fit kernels,
# fuzz with the riding-pick forms, HSA_ENABLE_INTERRUPT=0 on the driver-form bench line.
TAG=${1:-r3j}; REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
W2=$((1<<21)); W4=$((2<<21)); G4=$((1<<23)); G8=$((2<<23)); GF=$((3<<23))
stamp "best-fit A/B (bindings-only, C5 shard): shipped / wide 2 / wide 4 / grid:4 / grid:8 / 2048 blocks / wide4+grid:4 / wide4+grid:8 / wide4+2048 / wide2+grid:4"
timeout 600 python tools/bestfit_ab.py 0 $W2 $W4 $G4 $G8 $GF $((W4|G4)) $((W4|G8)) $((W4|GF)) $((W2|G4)) 0 2>&1 | tee $OUT/bestfit_ab.txt | tail -14
stamp "rocprofv3 split: shipped, wide4+grid:4"
cd /tmp
for v in 0 $((W4|G4)); do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bf_$v -o bf -- python $REPO/tools/bestfit_ab.py --no-oracle --steps 100 $v > $OUT/prof_bf_$v.log 2>&1
  f=$(find $OUT/prof_bf_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp $f $OUT/prof_bf_${v}_kernel_stats.csv; grep -i "bestfit" $f | cut -c1-160; }
done
stamp "counters of the best-fit kernels (separate passes)"
rocprofv3 -L > $OUT/counters_avail.txt 2>&1 || true
pmc() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o p -- python $REPO/tools/bestfit_ab.py --no-oracle --steps 10 0 > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVES
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pmc tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pmc ta TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
python - <<PY
import csv, glob, collections
for name in ("sq1","sq2","tcp","tcc","ta"):
    fs = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % name, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in fs:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]][r.get("Dispatch_Id","0")] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "bestfit" in k:
            print(name, k, {c: round(sum(v.values())/len(v)) for c, v in d.items()})
PY
cd $REPO
stamp "fuzz 150 s with the riding-pick forms"
timeout 400 python tools/fuzz_parity.py 150 7 > $OUT/fuzz.txt 2>&1; tail -3 $OUT/fuzz.txt
stamp "driver-form bench line: default env / HSA_ENABLE_INTERRUPT=0 (alternating, 2 each)"
for i in 1 2; do
  for e in "" "HSA_ENABLE_INTERRUPT=0"; do
    env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others --no-strong-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('env[$e] ms_per_step %.5f repeats %s' % (d['ms_per_step'], d.get('repeat_ms_per_step')))"
  done
done
stamp "done"
