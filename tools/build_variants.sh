#!/bin/bash
# Build-time variants of the fused kernel for A/B timing on the GPU box (kernels_fused.hpp: KSCHED_SPLIT_STAGE,
# KSCHED_STORE_POLICY, KSCHED_STAGGER, KSCHED_FUSED_THREADS).  Every variant is a complete, valid library; select one
# with KSCHED_LIB=<path>.
#   usage: bash tools/build_variants.sh name=FLAGS ...     e.g.  s0p2="-DKSCHED_SPLIT_STAGE=0 -DKSCHED_STORE_POLICY=2"
#   -> build/variants/libksched_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
n=0
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  out=build/variants/libksched_hip_${name}.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed \
      $flags -shared -o $out kube_scheduler_rs_reference_amd/csrc/ksched_api.hip &
  n=$((n+1)); if [ $((n % 8)) -eq 0 ]; then wait; fi
done; wait
ls build/variants
