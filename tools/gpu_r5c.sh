#!/bin/bash
# round 5, session C: full GPU suite (new: exchange with n > 1 through the stand-in, k-stream pipe), bench default + driver form with rotated inputs
OUT=$PWD/gpurun_out/r5c; mkdir -p $OUT; export TMPDIR=/tmp
echo "== sharded_rccl direct"; KSCHED_TEST_HOOKS=1 KSCHED_RCCL_LIB=$PWD/tests/cpp/libfake_rccl.so timeout 300 tests/cpp/host_tests sharded_rccl 2>&1 | tail -12 | tee $OUT/host_tests_sharded_rccl.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $OUT/pytest_gpu.log
echo "== bench driver form"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo rc=$?; tail -c 600 $OUT/bench_driver_form.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_driver_form.json") if l.startswith("{")][-1]); r=d["roofline"]; c=d["config"]
print("driver form: value %.3e step %.2f us kernel %.2f us frac %.3f step_frac %.3f repeats %s" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], c["step_frac_of_hbm_peak"], [round(x*1e3,2) for x in c["repeat_ms_per_step"]]))
print("parity", d["parity_check"]); print("input_rotation", c["input_rotation"]["batches"], c["input_rotation"]["bytes_resident"]); print("two_batches", c["two_batches_in_flight"])
PY
echo "== bench default"; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo rc=$?
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1]); r=d["roofline"]; c=d["config"]
print("default: value %.3e step %.2f us kernel %.2f us frac %.3f step_frac %.3f" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], c["step_frac_of_hbm_peak"]))
print("parity", d["parity_check"]); print({k:(v.get("ms_per_step"), v.get("mask_kernel_us")) for k,v in c["other_workloads"].items()})
PY
echo "== bench default, one input batch (A/B of the input rotation)"; timeout 600 python bench.py --input-batches 1 --no-others --no-cpu-baseline > $OUT/bench_default_one_input.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_default_one_input.json") if l.startswith("{")][-1]); r=d["roofline"]
print("one input batch: step %.2f us kernel %.2f us" % (d["ms_per_step"]*1e3, r["avg_kernel_us"]))
PY
