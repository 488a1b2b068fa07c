#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_pmc.sh <tag> "<COUNTER [COUNTER...]>" <bench args...>
# One counter group per call (separate passes, as MI355X_MICROARCH.md prescribes); summaries land in
# gpurun_out/pmc_<tag>_by_kernel.csv.  Useful groups:
#   "FETCH_SIZE"   "WRITE_SIZE"                         HBM-side traffic (FETCH_SIZE x2 on gfx950)
#   "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"    MFMA pipe occupancy
#   "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"   where waves wait
tag=$1; counters=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/pmc_$tag -- \
    python $R/bench.py "$@" --no-kernel-timing --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py "$f" gpurun_out/pmc_${tag}_by_kernel.csv
find gpurun_out/pmc_$tag -name "*.csv" -size +2M -delete   # keep gpurun_out small
head -5 gpurun_out/pmc_${tag}_by_kernel.csv 2>/dev/null | cut -c1-200
