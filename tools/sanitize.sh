#!/bin/bash
# Host-side sanitizer runs of the C++ mirror's tests (make sanitize builds build/san/host_tests_{asan,tsan}): AddressSanitizer + UBSan, and ThreadSanitizer,
# over HOST code only -- the worker pool, the staged snapshot update beside the POSTs, the draws' thread, the reaper, PodBatcher / run_batches, the sharded
# context.  Device code is not instrumented (GPU sanitizers are not available on this pool).
# (on a GPU box through gpurun: build/san/ is listed in .gpurunignore -- comment that line out for the session, or the binaries do not travel)
# usage: bash tools/sanitize.sh <outdir> [modes...]      modes: cpu (no GPU needed) gpu sharded comm sharded_rccl gpu3 (= gpu through a three-way shard over the RCCL
#                                                        stand-in), loop (tools/host_loop.py through build/san/objects_eval_*: a 20 000 x 2 000 and a C3-size
#                                                        100 000 x 5 000 batch, objects -> reconcile_batch -> POST sink -> staged snapshot update, batch + sequential);
#                                                        default: cpu, + the device halves when a GPU is visible
OUT=${1:-gpurun_out/san}; shift
mkdir -p $OUT
MODES=${@:-cpu}
if [ "$MODES" = "cpu" ] && command -v rocminfo >/dev/null && rocminfo 2>/dev/null | grep -q gfx950; then MODES="cpu gpu sharded comm sharded_rccl gpu3 loop"; fi
# leaks are checked in every mode; the ROCm runtime's own start-up allocations are suppressed by library name (tools/lsan.supp)
export ASAN_OPTIONS=detect_leaks=1 LSAN_OPTIONS=suppressions=$PWD/tools/lsan.supp:print_suppressions=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4 suppressions=$PWD/tools/tsan.supp"
rc_all=0
for san in ${SANS:-asan tsan}; do
  bin=build/san/host_tests_$san
  [ -x $bin ] || { echo "$bin missing (make sanitize)"; rc_all=1; continue; }
  for m in $MODES; do
    log=$OUT/${san}_$m.log
    # the stand-in for librccl and the k-replica shard only where the mode is about them (tests/test_host_mirror.py sets the same)
    hooks=""; arg=$m
    [ $m = sharded_rccl ] && hooks="KSCHED_TEST_HOOKS=1 KSCHED_RCCL_LIB=$PWD/tests/cpp/libfake_rccl.so"
    [ $m = gpu3 ] && hooks="KSCHED_TEST_HOOKS=1 KSCHED_RCCL_LIB=$PWD/tests/cpp/libfake_rccl.so KSCHED_SHARDED=3" && arg=gpu
    # ThreadSanitizer wants its own address-space layout: without address-space randomisation (setarch -R) where the kernel's is too wide for it
    pre=""; [ $san = tsan ] && setarch $(uname -m) -R true 2>/dev/null && pre="setarch $(uname -m) -R"
    if [ $m = loop ]; then
      env OBJECTS_EVAL_BIN=$PWD/build/san/objects_eval_$san LD_LIBRARY_PATH=$PWD/tests/cpp/hooks timeout 1500 $pre python tools/host_loop.py --sizes ${LOOP_SIZES:-20000x2000,100000x5000} --reps 2 > $log 2>&1; rc=$?
      grep -q FAILED $log && rc=1
      n_asan=$(grep -c "ERROR: AddressSanitizer" $log); n_ub=$(grep -c "runtime error:" $log); n_tsan=$(grep -c "WARNING: ThreadSanitizer" $log)
      echo "$san $m: exit $rc | $(grep -c " bound in " $log) batch form(s) ran | AddressSanitizer errors $n_asan, UBSan reports $n_ub, ThreadSanitizer reports $n_tsan"
      [ $rc -ne 0 ] && rc_all=1
      continue
    fi
    env $hooks timeout 900 $pre $bin $arg > $log 2>&1; rc=$?
    n_asan=$(grep -c "ERROR: AddressSanitizer" $log); n_ub=$(grep -c "runtime error:" $log); n_tsan=$(grep -c "WARNING: ThreadSanitizer" $log)
    echo "$san $m: exit $rc | $(grep -E '^[0-9]+ test\(s\)' $log | tail -1) | AddressSanitizer errors $n_asan, UBSan reports $n_ub, ThreadSanitizer reports $n_tsan"
    [ $rc -ne 0 ] && rc_all=1
  done
done | tee $OUT/summary.txt
exit $rc_all
