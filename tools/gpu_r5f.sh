#!/bin/bash
# round 5, session F: prefetch behind the barrier: trace + bench A/B; full GPU suite
OUT=$PWD/gpurun_out/r5f; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]; c=d["config"]
print("%-46s step %.2f us kernel %.2f us frac %.3f step_frac %.3f repeats %s" % (sys.argv[2], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], c["step_frac_of_hbm_peak"], [round(x*1e3,2) for x in c["repeat_ms_per_step"]]))
PY
}
echo "== trace, rotated"; timeout 300 python tools/trace_fused.py --workload C3 --pick --rotate 6 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_C3_rotated.txt | sed -n 2,16p
echo "== trace, one batch"; timeout 300 python tools/trace_fused.py --workload C3 --pick 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_C3_one.txt | sed -n 2,16p
B="--no-cpu-baseline --no-others"
timeout 300 python bench.py $B > $OUT/a.json 2>/dev/null; summ $OUT/a.json "default (6 input batches, prefetch)"
timeout 300 python bench.py $B --debug 536870912 --no-parity-check > $OUT/b.json 2>/dev/null; summ $OUT/b.json "6 input batches, NO prefetch (debug bit 29)"
timeout 300 python bench.py $B --input-batches 1 > $OUT/c.json 2>/dev/null; summ $OUT/c.json "1 input batch, prefetch"
timeout 300 python bench.py $B --steps 20 --warmup 5 > $OUT/e.json 2>/dev/null; summ $OUT/e.json "driver form (6 input batches, prefetch)"
timeout 300 python bench.py $B --workload C4s > $OUT/f.json 2>/dev/null; summ $OUT/f.json "C4s prefetch"
timeout 300 python bench.py $B --workload C5s --steps 300 > $OUT/h.json 2>/dev/null; summ $OUT/h.json "C5s"
timeout 300 python bench.py $B --workload C2 > $OUT/i.json 2>/dev/null; summ $OUT/i.json "C2"
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee $OUT/pytest_gpu.log
