#!/bin/bash
# ONE command that pins the ksched evaluator's resource-fit and selector results to the reference's own predicates:
#
#     rust/pin_parity.sh /path/to/kube-scheduler-rs-reference
#
# Needs cargo + network (or a vendored registry) for the reference's dependencies; needs NO GPU and NO ksched library.
# It patches a copy of the reference (rust/apply.sh), runs the reference's `fits` (pure half of can_pod_fit) and
# `does_node_selector_match` on tests/golden/*_objects.json, writes tests/golden/ref_*.json, then runs the comparison.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd); REPO=$(dirname "$HERE")
WORK=${KSCHED_PIN_WORKDIR:-$(mktemp -d)}
"$HERE/apply.sh" "$1" "$WORK/scheduler-v0"
cd "$WORK/scheduler-v0"
KSCHED_GOLDEN_DIR="$REPO/tests/golden" cargo test --release parity_dump -- --nocapture
cd "$REPO"
python -m pytest tests/test_reference_fixtures.py -q -rsx
# with an MI355X and the built library, additionally:  (cd $WORK/scheduler-v0 && KSCHED_LIB_DIR=$REPO/kube_scheduler_rs_reference_amd \
#   KSCHED_GOLDEN_DIR=$REPO/tests/golden cargo test --release --features ksched device_parity -- --nocapture)
