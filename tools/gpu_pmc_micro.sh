#!/bin/bash
# usage (via gpurun): tools/gpu_pmc_micro.sh <tag> "<COUNTERS>" <python script + args>
# PMC pass over a microbenchmark script (not bench.py); summary -> gpurun_out/pmc_<tag>_by_kernel.csv
tag=$1; counters=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/pmc_$tag -- python "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py "$f" gpurun_out/pmc_${tag}_by_kernel.csv
rm -rf gpurun_out/pmc_$tag
