#!/bin/bash
# round 5, session E: wide pods / WARN lines / ksched_pick through the full GPU suite; phase trace with rotated inputs; bench lines
OUT=$PWD/gpurun_out/r5e; mkdir -p $OUT; export TMPDIR=/tmp
echo "== host_tests gpu"; timeout 600 tests/cpp/host_tests gpu 2>&1 | grep -v "^ok " | tail -20 | tee $OUT/host_tests_gpu.txt
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $OUT/pytest_gpu.log
echo "== trace, rotated"; timeout 300 python tools/trace_fused.py --workload C3 --pick --rotate 6 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_C3_rotated.txt | head -22
echo "== trace, one batch"; timeout 300 python tools/trace_fused.py --workload C3 --pick 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_C3_one.txt | head -22
echo "== bench driver form"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo rc=$?
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_driver_form.json") if l.startswith("{")][-1]); r=d["roofline"]; c=d["config"]
print("driver form: value %.3e step %.2f us kernel %.2f us frac %.3f step_frac %.3f repeats %s" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], c["step_frac_of_hbm_peak"], [round(x*1e3,2) for x in c["repeat_ms_per_step"]]))
print("parity", d["parity_check"]["mismatches"], "others", {k:(round(v["ms_per_step"]*1e3,1), round(v["mask_kernel_us"],1)) for k,v in c["other_workloads"].items()}, "two", round(c["two_batches_in_flight"]["ms_per_step"]*1e3,2), "in_place", round(c["in_place"]["ms_per_step"]*1e3,2))
PY
