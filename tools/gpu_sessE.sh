cd $GRAFT_REPO_ROOT
for dbg in 0 2; do for wl in C3 C4s C5s; do timeout 300 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --debug $dbg 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('debug=$dbg $wl: step %.1f us kernel %.2f us frac %.3f' % (d['ms_per_step']*1e3, r['avg_kernel_us'], r['frac']))"; done; done
bash tools/gpu_pmc2.sh sessE_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" -- python $PWD/bench.py --workload C4s --steps 10 --warmup 2 --no-cpu-baseline
bash tools/gpu_pmc2.sh sessE_sq2 "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS" -- python $PWD/bench.py --workload C4s --steps 10 --warmup 2 --no-cpu-baseline
