#!/bin/bash
# round 5, session I: pick forms under the new regime (fresh inputs, nt stores); fuzz with the multi-device sequence; new tests
OUT=$PWD/gpurun_out/r5i; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for wl in C3 C4s; do for fp in 1 0 2 3; do
  timeout 300 python bench.py --workload $wl --fused-pick $fp --no-cpu-baseline --no-others --repeats 2 2>/dev/null | tail -1 > $OUT/pick_${wl}_$fp_$rep.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/pick_${wl}_$fp_$rep.json")); r=d["roofline"]; c=d["config"]
    print("$wl fused-pick=$fp rep $rep: step %.2f us kernel %.2f us pick=%s repeats %s parity %s" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], c["pick_launch"], [round(x*1e3,2) for x in c["repeat_ms_per_step"]], d["parity_check"]["mismatches"]))
except Exception as e:
    print("$wl $fp FAILED", e)
PY
done; done; done 2>&1 | tee $OUT/pick_forms.txt
echo "== fuzz with hooks (60 s)"; KSCHED_TEST_HOOKS=1 KSCHED_RCCL_LIB=$PWD/tests/cpp/libfake_rccl.so timeout 200 python tools/fuzz_parity.py 60 424242 2>&1 | tail -3 | tee $OUT/fuzz_hooks.txt
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_host_mirror.py tests/test_gpu_objects.py tests/test_gpu_fused_pick.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6 | tee $OUT/pytest_new.log
