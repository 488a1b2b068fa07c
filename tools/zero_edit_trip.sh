#!/bin/bash
# One gpurun call that exercises INTEGRATION.md section A on the MI355X: a SCRATCH copy of the
# reference's python package (never committed: .ref_scratch/ is git-ignored) travels with the
# snapshot, tests/test_gpu_zero_edit.py runs the reference's unmodified model files over the
# product mirrors, and the scratch copy is removed again.  Usage: tools/zero_edit_trip.sh [extra cmd]
set -e
cd "$(dirname "$0")/.."
REF=${PONDERV2_REFERENCE:-/root/reference}
rm -rf .ref_scratch && mkdir -p .ref_scratch
cp -r "$REF/ponder" "$REF/configs" .ref_scratch/
find .ref_scratch -name __pycache__ -type d -exec rm -rf {} +
trap 'rm -rf .ref_scratch' EXIT
EXTRA=${1:-true}
/usr/local/graft/bin/gpurun --timeout ${TRIP_TIMEOUT:-900} -- "
mkdir -p gpurun_out
export PONDERV2_REFERENCE=\$PWD/.ref_scratch
python -m pytest tests/test_gpu_zero_edit.py -m gpu -q -s -x 2>&1 | tee gpurun_out/zero_edit.txt
$EXTRA
"
