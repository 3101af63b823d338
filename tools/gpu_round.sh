#!/bin/bash
# One GPU-box session that produces the round's evidence set.  usage: bash tools/gpu_round.sh <tag> [steps...]
#   steps (default: test bench prof pmc lines trace smoke host): probe test smoke bench prof pmc lines ab dist evflags abpick host fuzz probe2 trace
# Everything lands under gpurun_out/<tag>/ ; tools/collect_profiles.py copies the summaries into profiles/.
TAG=${1:-r}; shift
STEPS=${@:-"test bench prof pmc lines trace smoke host"}
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
stamp() { echo "[$(date +%H:%M:%S)] $*"; }

if has probe; then
  stamp "toolchain probe (BASELINE.md section 4 asks whether the reference's own toolchain exists on the GPU box)"
  { for t in cargo rustc rustup go javac node gcc g++ hipcc python3; do printf "%-8s " $t; (command -v $t >/dev/null && ($t --version 2>&1 | head -1)) || echo MISSING; done
    echo "nproc $(nproc)"; ls -d ~/.cargo ~/.rustup 2>&1 | head -2; rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing"; } > $OUT/toolchain_probe.txt 2>&1
  cat $OUT/toolchain_probe.txt
fi
if has test; then
  stamp "pytest -m gpu"
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
fi
if has smoke; then
  stamp "smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
fi
if has bench; then
  stamp "bench (default command, and the driver's form: --steps 20 --warmup 5)"
  timeout 600 python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.log 2>&1; tail -1 $OUT/bench_driver_form.log > $OUT/bench_driver_form.json
  for f in bench_default bench_driver_form; do python - <<PY
import json
d=json.load(open("$OUT/$f.json")); r=d["roofline"]; c=d["cpu_baseline"]; g=d["config"]
print("$f: %.3e evals/s step %.2f us (step frac %.3f) kernel %.2f us %.0f GB/s frac %.3f pick=%s rot=%s | cpu %.3e (%d cores) single %.3e encoded %.3e" % (d["value"], d["ms_per_step"]*1e3, g["step_frac_of_hbm_peak"], r["avg_kernel_us"], r["achieved"], r["frac"], g["pick_launch"], g["mask_rotation"], c["value"], c["cores"], c["single_core_value"], c["encoded_loop_value"]))
for k in ("in_place", "two_batches_in_flight"):
    print("   ", k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in (g.get(k) or {}).items() if kk != "note"})
for k, v in (g.get("other_workloads") or {}).items():
    print("   ", k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("workload", "pick_alone_note")})
PY
  done
fi
if has prof; then
  stamp "rocprofv3 --kernel-trace --stats -- python bench.py"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -o r -- python $REPO/bench.py > $OUT/prof_default.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_driver_form -o r -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/prof_driver_form.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5s -o r -- python $REPO/bench.py --workload C5s --no-cpu-baseline --live-traffic off --no-others > $OUT/prof_c5s.log 2>&1
  cd $REPO
  grep "^{\"metric" $OUT/prof_default.log > $OUT/prof_default_bench.json
  grep "^{\"metric" $OUT/prof_driver_form.log > $OUT/prof_driver_form_bench.json
  f=$(find $OUT/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prof_default_kernel_stats.csv && head -6 $f | cut -c1-260
  f=$(find $OUT/prof_driver_form -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prof_driver_form_kernel_stats.csv && head -4 $f | cut -c1-260
  f=$(find $OUT/prof_c5s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prof_c5s_kernel_stats.csv && head -6 $f | cut -c1-200
  # the dominant kernel's durations by phase of the command (ramp / timed region / repeats / legs): what corresponds to roofline.avg_kernel_us
  for w in default driver_form; do
    t=$(find $OUT/prof_$w -name "*kernel_trace.csv" | head -1)
    [ -n "$t" ] && [ -s $OUT/prof_${w}_bench.json ] && python $REPO/tools/rocprof_timed_region.py $t $OUT/prof_${w}_bench.json "k_eval_fused" > $OUT/prof_${w}_by_phase.json 2>$OUT/prof_${w}_by_phase.err && python -c "
import json; d=json.load(open('$OUT/prof_${w}_by_phase.json')); s=d['segments']; print('$w by phase:', d['kernel'], {k: (v['launches'], round(v['mean_us'], 2)) for k, v in s.items()}, 'all', round(d['all_launches_mean_us'], 2))"
  done
  find $OUT -name "*.db" -size +2M -delete
fi
if has pmc; then
  stamp "PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE; calibration + bench)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $REPO/tools/pmc_calib $REPO/tools/pmc_calib.hip 2>&1 | tail -2
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o p -- $REPO/tools/pmc_calib > $OUT/calib_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o p -- $REPO/tools/pmc_calib > $OUT/calib_write.log 2>&1
  for wl in ${PMC_WLS:-C3 C4s}; do
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${wl}_fetch -o p -- python $REPO/bench.py --workload $wl --steps 10 --warmup 2 --ramp-ms 0 --kernel-samples 2 --no-cpu-baseline --live-traffic off > $OUT/${wl}_fetch.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${wl}_write -o p -- python $REPO/bench.py --workload $wl --steps 10 --warmup 2 --ramp-ms 0 --kernel-samples 2 --no-cpu-baseline --live-traffic off > $OUT/${wl}_write.log 2>&1
  done
  # a third pass: SQ activity of the mask kernel (VALU / LDS / wait split) on the default workload
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/C3_sq -o p -- python $REPO/bench.py --steps 10 --warmup 2 --ramp-ms 0 --kernel-samples 2 --no-cpu-baseline --live-traffic off > $OUT/C3_sq.log 2>&1
  cd $REPO
  python tools/pmc_traffic.py $OUT ${PMC_WLS:-C3 C4s} > $OUT/pmc_traffic.log 2>&1; tail -30 $OUT/pmc_traffic.log
  # drop the bulky raw traces, keep the counter csvs
  find $OUT -name "*kernel_trace.csv" -size +2M -delete
fi
if has lines; then
  stamp "bench lines of the other workloads"
  for wl in ${LINE_WLS:-C2 C3h C4s C5s}; do
    timeout 600 python bench.py --workload $wl --no-cpu-baseline --live-traffic off 2>&1 | tail -1 > $OUT/bench_${wl}.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${wl}.json")); r=d["roofline"]
    print("$wl: %.3e evals/s  step %.1f us  kernel %.1f us  %.0f GB/s  frac %.3f" % (d["value"], d["ms_per_step"]*1e3, r["avg_kernel_us"], r["achieved"], r["frac"]))
except Exception as e:
    print("$wl: FAILED", e)
PY
  done
fi
if has ab; then
  stamp "A/B: bench with kernel debug bits ${AB_BITS:-64} (results of a debug run are not valid outputs; timing only)"
  for wl in ${AB_WLS:-C3 C4s}; do for bits in ${AB_BITS:-64}; do
    timeout 600 python bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline --live-traffic off --debug $bits 2>&1 | tail -1 > $OUT/ab_${wl}_${bits}.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_${wl}_${bits}.json")); r=d["roofline"]
    print("$wl debug=$bits: step %.1f us  kernel %.1f us  frac %.3f" % (d["ms_per_step"]*1e3, r["avg_kernel_us"], r["frac"]))
except Exception as e:
    print("$wl debug=$bits: FAILED", e)
PY
  done; done
fi
if has dist; then
  stamp "N > 1 bench path in a ONE-rank RCCL group (KSCHED_BENCH_FORCE_DIST=1): C ABI communicator vs torch, pipe vs one stream"
  KSCHED_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --live-traffic off 2>$OUT/dist_default.err | tail -1 > $OUT/dist_default.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/dist_default.json")); c=d["config"]
    print("default N>1 path (one rank): %s | gather/step %.1f us/step | no gather %s | eff vs no gather %s | strong leg %s" % (c["workload"][:24], d["ms_per_step"]*1e3, (c.get("no_allgather") or {}).get("ms_per_step"), c.get("scaling_efficiency_vs_no_allgather"), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (c.get("configs3_strong") or {}).items() if k not in ("workload", "no_allgather")}))
except Exception as e:
    print("default N>1 path FAILED", e); print(open("$OUT/dist_default.err").read()[-1500:])
PY
  for wl in C3 C4s; do for mode in "" "--torch-gather" "--one-stream"; do
    KSCHED_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload $wl --no-cpu-baseline --live-traffic off $mode 2>&1 | tail -1 > $OUT/dist_${wl}_${mode#--}.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/dist_${wl}_${mode#--}.json")); c=d["config"]
    print("$wl $mode: gather/step %.1f us/step %.3e evals/s | other gather cadence: %s | no gather: %s | mask kernel %.1f us" % (d["ms_per_step"]*1e3, d["value"], (c.get("allgather_every_step") or c.get("allgather_every_4") or {}).get("ms_per_step"), (c.get("no_allgather") or {}).get("ms_per_step"), d["roofline"]["avg_kernel_us"]))
except Exception as e:
    print("$wl $mode FAILED", e)
PY
  done; done
fi
if has evflags; then
  stamp "A/B of the HIP event flags of the per-dispatch kernel timing (bench.py post-pass) vs rocprofv3 (the prof step)"
  for f in 0x0 0x20000000 0x40000000; do
    KSCHED_TIMING_EVENT_FLAGS=$f timeout 300 python bench.py --no-cpu-baseline --live-traffic off 2>&1 | tail -1 > $OUT/evflags_$f.json
    python -c "
import json; r=json.load(open('$OUT/evflags_$f.json'))['roofline']; print('flags $f: mean %.2f median %.2f min %.2f max %.2f us' % (r['avg_kernel_us'], r['median_kernel_us'], r['min_kernel_us'], r['max_kernel_us']))"
  done
fi
if has abpick; then
  stamp "A/B of the picks (KSCHED_OPT_DEBUG): sampled pick eager draws (bits 8-9), best-fit one stage (bit 10) / lane words (bits 12-15); bindings-only steps"
  for wl in C3 C2; do for dbg in 0 256 512 768; do
    timeout 200 python bench.py --workload $wl --no-cpu-baseline --live-traffic off --no-mask --debug $dbg 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl debug=$dbg: bindings-only step %.2f us' % (d['ms_per_step']*1e3))"
  done; done
  for dbg in 1024 4096 8192 16384 32768; do
    timeout 200 python bench.py --workload C5s --no-cpu-baseline --live-traffic off --no-mask --debug $dbg --steps 300 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C5s debug=$dbg: bindings-only step %.1f us' % (d['ms_per_step']*1e3))"
  done
fi
if has host; then
  stamp "host-side cost of the snapshot calls"
  timeout 300 python tools/host_costs.py > $OUT/host_costs.txt 2>&1; cat $OUT/host_costs.txt
fi
if has hostloop; then
  stamp "objects -> reconcile_batch through the C++ host mirror (C3-size batch and the round-3 sizes), phases of the last repeat"
  KSCHED_HOST_TIMING=2 timeout 600 python tools/host_loop.py --sizes 100000x5000 --modes batch --reps 5 > $OUT/host_loop_C3.txt 2>&1; cat $OUT/host_loop_C3.txt | cut -c1-260
  timeout 600 python tools/host_loop.py --sizes 5000x500,20000x2000 --reps 3 > $OUT/host_loop_small.txt 2>&1; grep -v "phase\|reconcile_batch" $OUT/host_loop_small.txt | cut -c1-260
fi
if has spread; then
  stamp "the default line in ${SPREAD_N:-5} fresh processes (VERDICT r5 item 1: C3 per-step spread)"
  for i in $(seq 1 ${SPREAD_N:-5}); do
    timeout 600 python bench.py --no-cpu-baseline --live-traffic off --no-others --repeats 1 2>/dev/null | tail -1 > $OUT/spread_$i.json
  done
  python - <<PY
import json
v = [json.load(open("$OUT/spread_%d.json" % i)) for i in range(1, ${SPREAD_N:-5} + 1)]
st = [d["ms_per_step"] * 1e3 for d in v]; ke = [d["roofline"]["avg_kernel_us"] for d in v]
print("C3 step us in fresh processes: " + " ".join("%.2f" % x for x in st) + "  -> spread (max - min) / min = %.1f %%" % ((max(st) - min(st)) / min(st) * 100))
print("C3 mask kernel us:             " + " ".join("%.2f" % x for x in ke) + "  -> spread %.1f %%" % ((max(ke) - min(ke)) / min(ke) * 100))
json.dump({"step_us": st, "kernel_us": ke, "step_spread_frac": (max(st) - min(st)) / min(st), "kernel_spread_frac": (max(ke) - min(ke)) / min(ke)}, open("$OUT/spread.json", "w"))
PY
  for i in 1 2 3; do timeout 600 python bench.py --workload C5s --no-cpu-baseline --live-traffic off --steps 300 --repeats 1 2>/dev/null | tail -1 > $OUT/spread_C5s_$i.json; done
  python - <<PY
import json
v = [json.load(open("$OUT/spread_C5s_%d.json" % i)) for i in (1, 2, 3)]
print("C5 shard in 3 fresh processes (masks from ksched_mask_alloc: probe-and-keep): mask kernel us " + " ".join("%.1f" % d["roofline"]["avg_kernel_us"] for d in v) + " | step us " + " ".join("%.1f" % (d["ms_per_step"] * 1e3) for d in v))
PY
fi
if has box; then
  bash tools/box_fingerprint.sh > $OUT/box.txt 2>&1; grep -i "unique\|nproc" $OUT/box.txt
fi
if has fuzz; then
  stamp "randomised differential parity against the oracle (${FUZZ_S:-60} s, seed ${FUZZ_SEED:-20260923})"
  timeout $(( ${FUZZ_S:-60} + 60 )) python tools/fuzz_parity.py ${FUZZ_S:-60} ${FUZZ_SEED:-20260923} > $OUT/fuzz.txt 2>&1; tail -3 $OUT/fuzz.txt
fi
if has probe2; then
  stamp "update + pick loop per context, consecutive loops (tools/stream_probe.py --repeat)"
  PROBE_N=2 timeout 200 python tools/stream_probe.py --repeat 2>&1 | grep "context #\|consecutive" > $OUT/stream_probe.txt; cat $OUT/stream_probe.txt
fi
if has trace; then
  stamp "fused kernel phase trace (C3), with the riding pick and without"
  timeout 300 python tools/trace_fused.py --workload C3 --pick > $OUT/trace_C3.txt 2>&1; head -16 $OUT/trace_C3.txt
  timeout 300 python tools/trace_fused.py --workload C3 > $OUT/trace_C3_nopick.txt 2>&1; head -15 $OUT/trace_C3_nopick.txt
fi
stamp "done"
