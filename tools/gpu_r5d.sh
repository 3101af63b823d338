#!/bin/bash
# round 5, session D: operand prefetch A/B with rotated inputs; full GPU suite
OUT=$PWD/gpurun_out/r5d; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]; c=d["config"]
print("%-46s step %.2f us kernel %.2f us frac %.3f step_frac %.3f repeats %s parity %s" % (sys.argv[2], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"], c["step_frac_of_hbm_peak"], [round(x*1e3,2) for x in c["repeat_ms_per_step"]], d["parity_check"]["mismatches"]))
o=c.get("other_workloads")
if o: print("   others:", {k:(round(v["ms_per_step"]*1e3,1), round(v["mask_kernel_us"],1)) for k,v in o.items()})
PY
}
B="--no-cpu-baseline --no-others"
timeout 300 python bench.py $B > $OUT/a.json 2>/dev/null; summ $OUT/a.json "default (6 input batches, prefetch)"
timeout 300 python bench.py $B --debug 536870912 --no-parity-check > $OUT/b.json 2>/dev/null; summ $OUT/b.json "6 input batches, NO prefetch (debug bit 29)"
timeout 300 python bench.py $B --input-batches 1 > $OUT/c.json 2>/dev/null; summ $OUT/c.json "1 input batch, prefetch"
timeout 300 python bench.py $B --input-batches 1 --debug 536870912 --no-parity-check > $OUT/d.json 2>/dev/null; summ $OUT/d.json "1 input batch, NO prefetch"
timeout 300 python bench.py $B --steps 20 --warmup 5 > $OUT/e.json 2>/dev/null; summ $OUT/e.json "driver form (6 input batches, prefetch)"
timeout 300 python bench.py $B --workload C4s > $OUT/f.json 2>/dev/null; summ $OUT/f.json "C4s prefetch"
timeout 300 python bench.py $B --workload C4s --debug 536870912 --no-parity-check > $OUT/g.json 2>/dev/null; summ $OUT/g.json "C4s NO prefetch"
timeout 300 python bench.py $B --workload C5s --steps 300 > $OUT/h.json 2>/dev/null; summ $OUT/h.json "C5s prefetch"
timeout 300 python bench.py $B --workload C5s --steps 300 --debug 536870912 --no-parity-check > $OUT/i.json 2>/dev/null; summ $OUT/i.json "C5s NO prefetch"
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $OUT/pytest_gpu.log
