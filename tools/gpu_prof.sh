#!/bin/bash
# usage: tools/gpu_prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/**/kernel_stats.csv
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python $R/bench.py "$@" --no-kernel-timing --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${tag}_kernel_stats.csv \;
