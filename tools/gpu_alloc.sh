#!/bin/bash
# the two allocation speeds -- survey of the allocation paths + counter passes.  usage: bash tools/gpu_alloc.sh <tag> [steps]
TAG=${1:-alloc}; shift
STEPS=${@:-"list survey pmc"}
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export KSCHED_LIB=$REPO/tests/cpp/hooks/libksched_hip.so  # the measurement allocation paths exist in the test build of the library only
has() { [[ " $STEPS " == *" $1 "* ]]; }
stamp() { echo "[$(date +%H:%M:%S)] $*"; }
if has list; then
  stamp "counters rocprofv3 lists on this box"
  (cd /tmp && timeout 120 rocprofv3 -L > $OUT/rocprofv3_list_avail.txt 2>&1); grep -c "" $OUT/rocprofv3_list_avail.txt
  grep -o "UTCL[0-9A-Z_a-z]*\|TCC_EA0_WRREQ[0-9A-Z_a-z]*" $OUT/rocprofv3_list_avail.txt | sort -u | tr '\n' ' ' | cut -c1-1500
fi
if has test; then
  stamp "pytest -m gpu"
  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
fi
if has survey; then
  for wl in ${SURVEY_WLS:-C5s C3 C4s}; do
    stamp "survey $wl"
    timeout 600 python tools/alloc_probe.py $wl survey k=${SURVEY_K:-4} hows=${SURVEY_HOWS:-1,2,3,4,5,7} > $OUT/survey_$wl.txt 2>&1; grep "^==\|failed\|Error" $OUT/survey_$wl.txt | head -20
  done
fi
if has pmc; then
  i=0
  for set in ${PMC_SETS:-"TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
             "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum" \
             "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
             "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
             "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL" \
             "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_IB_STALL_sum"}; do
    i=$((i+1))
    stamp "pmc pass $i: $set"
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -o p -- python $REPO/tools/alloc_probe.py ${PMC_WL:-C5s} pmc k=${PMC_K:-5} hows=${PMC_HOWS:-1,2} n=${PMC_N:-12} > $OUT/pmc_$i.log 2>&1)
    find $OUT/pmc_$i -name "*kernel_trace.csv" -delete; find $OUT/pmc_$i -name "*.db" -delete
    python tools/alloc_pmc_summary.py $OUT/pmc_$i.log $OUT/pmc_$i > $OUT/pmc_${i}_summary.txt 2>&1; grep "corr with time" $OUT/pmc_${i}_summary.txt | cut -c1-170
  done
fi
if has debugfs; then
  stamp "can the physical placement be seen? (debugfs)"
  { mount | grep -i debug; ls /sys/kernel/debug 2>&1 | head; ls /sys/kernel/debug/dri 2>&1 | head; for f in /sys/kernel/debug/dri/*/amdgpu_vram_mm; do echo "== $f"; head -40 $f; done; } > $OUT/debugfs.txt 2>&1; head -12 $OUT/debugfs.txt
fi
if has repeat; then
  for rep in ${REPEAT_IDS:-1 2 3}; do for wl in ${REPEAT_WLS:-C5s}; do
    stamp "fresh process $rep: survey $wl, passes=${REPEAT_PASSES:-4}"
    timeout 600 python tools/alloc_probe.py $wl survey k=${REPEAT_K:-4} hows=${REPEAT_HOWS:-1,4,7,2} passes=${REPEAT_PASSES:-4} > $OUT/repeat_${wl}_$rep.txt 2>&1; grep "^==\|failed\|Error" $OUT/repeat_${wl}_$rep.txt | head -20
  done; done
fi
if has pitch; then
  for wl in ${PITCH_WLS:-C5s C4s C3}; do
    stamp "row pitch sweep $wl"
    timeout 600 python tools/alloc_probe.py $wl pitch k=${PITCH_K:-3} passes=2 > $OUT/pitch_$wl.txt 2>&1; grep "^pitch\|Error" $OUT/pitch_$wl.txt | cut -c1-260
  done
fi
stamp done
