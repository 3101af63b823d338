OUT=gpurun_out/r6r; mkdir -p $OUT
bash tools/box_fingerprint.sh > $OUT/box.txt 2>&1; grep -i "unique\|nproc" $OUT/box.txt
timeout 900 python -m pytest tests/test_host_mirror.py tests/test_gpu_objects.py -m gpu -x -q > $OUT/pytest_host.log 2>&1; grep -n "passed\|failed\|Error" $OUT/pytest_host.log | tail -5
KSCHED_HOST_TIMING=2 python tools/host_loop.py --sizes 100000x5000 --modes batch --reps 5 > $OUT/host_loop_c3_phases.txt 2>&1; cat $OUT/host_loop_c3_phases.txt | cut -c1-300
python tools/host_loop.py --sizes 100000x5000 --modes batch --reps 3 --warn > $OUT/host_loop_c3_warn.txt 2>&1; grep -v phase $OUT/host_loop_c3_warn.txt | cut -c1-300
timeout 900 python bench.py --steps 200 --live-traffic off > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench.json
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print(json.dumps(d["config"]["end_to_end"], indent=1)); print(d["ms_per_step"]*1e3, d["roofline"]["frac"])
PY
