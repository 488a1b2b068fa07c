#!/bin/bash
# usage (via gpurun): tools/gpu_miopen_search.sh <bench args...>
# Runs MIOpen's solver search for the dense-conv shapes of a bench configuration and leaves the
# merged user find-db / kernel cache under gpurun_out/miopen_cache (copy it over miopen_cache/).
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/miopen_cache && cp -r $R/miopen_cache/* $R/gpurun_out/miopen_cache/
export MIOPEN_USER_DB_PATH=$R/gpurun_out/miopen_cache/db MIOPEN_CUSTOM_CACHE_DIR=$R/gpurun_out/miopen_cache/cache
PV2_MIOPEN_SEARCH=1 timeout 900 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" > $R/gpurun_out/miopen_search.json 2> $R/gpurun_out/miopen_search.err
echo "search rc=$? $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/miopen_search.json)"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > $R/gpurun_out/miopen_after.json 2>/dev/null
echo "after rc=$? $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/miopen_after.json) $(grep -o '"final_loss": [0-9.e-]*' $R/gpurun_out/miopen_after.json)"
du -sh $R/gpurun_out/miopen_cache; ls -la $R/gpurun_out/miopen_cache/db
