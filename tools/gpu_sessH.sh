cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/sessH_pytest.log 2>&1; tail -5 gpurun_out/sessH_pytest.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --depth 2 2>&1 | tail -1 | cut -c1-400
