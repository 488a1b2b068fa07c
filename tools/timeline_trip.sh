#!/bin/bash
# Device timeline of the default step (two streams / one): gpurun_out/timeline_<tag>.txt
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
for mode in ${MODES:-two one}; do
  cd /tmp && export TMPDIR=/tmp
  if [ $mode = one ]; then export PV2_WGRAD_STREAM=0; else unset PV2_WGRAD_STREAM; fi
  rm -rf /tmp/tl_$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$mode -- python $R/bench.py --steps 8 --warmup 4 --no-kernel-timing --no-cpu-baseline "$@" > $O/timeline_$mode.log 2>&1
  cd $R
  f=$(find /tmp/tl_$mode -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $f 5 > $O/timeline_$mode.txt 2>&1
  head -70 $O/timeline_$mode.txt
done
