cd $GRAFT_REPO_ROOT
for wl in C4s C5s; do timeout 200 python tools/trace_fused.py --workload $wl 2>&1 | grep -v amdgpu.ids | grep -v "median by tile"; done
