#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 10 --warmup 3 --no-kernel-timing --no-cpu-baseline > $O/stream_trip.log 2>&1
cd $R
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python tools/stream_breakdown.py $f 13 > $O/stream_breakdown.txt 2>&1
head -120 $O/stream_breakdown.txt
