#!/bin/bash
# usage: rust/apply.sh <reference checkout> <output dir>
# Copies the reference (acrlabs/kube-scheduler-rs-reference) to <output dir> and overlays the ksched binding on it:
#   patches/0001  src/predicates.rs : pure `fits()` seam under can_pod_fit (arithmetic untouched), check_node_validity_batch
#   patches/0002  src/main.rs       : feature ksched: select_node_for_pod -> the batch task (ready_chunks -> one device call), the pod watch
#   patches/0004  src/util.rs       : Context gains `picker`, the channel to the batch task (feature ksched)
#   patches/0003  Cargo.toml        : feature `ksched`, dev-dependencies serde / serde_json
#   build.rs, src/ksched_sys.rs, src/ksched.rs, src/predicates/{parity_dump,device_parity}.rs   (new files)
set -euo pipefail
REF=$(cd "$1" && pwd); OUT=$2; HERE=$(cd "$(dirname "$0")" && pwd)
rm -rf "$OUT"; mkdir -p "$OUT"; cp -r "$REF"/. "$OUT"/; rm -rf "$OUT/.git" "$OUT/target"
cd "$OUT"
for p in "$HERE"/patches/*.patch; do patch -p1 --no-backup-if-mismatch < "$p"; done
cp "$HERE/build.rs" .
cp "$HERE/src/ksched_sys.rs" "$HERE/src/ksched.rs" src/
cp "$HERE/src/predicates/parity_dump.rs" "$HERE/src/predicates/device_parity.rs" src/predicates/
echo "overlay applied in $OUT"
