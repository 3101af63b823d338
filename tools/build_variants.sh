#!/bin/bash
# Build-time variants of the fused kernel for A/B timing on the GPU box (kernels_fused.hpp: KSCHED_SPLIT_STAGE,
# KSCHED_STORE_POLICY).  Every variant is a complete, valid library; select one with KSCHED_LIB=<path>.
#   usage: bash tools/build_variants.sh   -> build/variants/libksched_hip_<split><policy>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
for split in 0 1; do for pol in 0 1 2 3; do
  out=build/variants/libksched_hip_s${split}p${pol}.so
  if [ ! -f $out ] || [ kube_scheduler_rs_reference_amd/csrc/kernels_fused.hpp -nt $out ] || [ kube_scheduler_rs_reference_amd/csrc/ksched_api.hip -nt $out ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed \
      -DKSCHED_SPLIT_STAGE=$split -DKSCHED_STORE_POLICY=$pol -shared -o $out kube_scheduler_rs_reference_amd/csrc/ksched_api.hip &
  fi
done; wait; done
ls -la build/variants
