#!/bin/bash
# usage: tools/r06_prof.sh <tag>  -> gpurun_out/r06/kernel_stats_<tag>.csv (+ single-stream), breakdown
set -u
tag=$1
O=gpurun_out/r06; mkdir -p $O
bash tools/gpu_prof.sh r06$tag --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r06${tag}_kernel_stats.csv $O/kernel_stats_$tag.csv
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh r06${tag}s --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r06${tag}s_kernel_stats.csv $O/kernel_stats_${tag}_single_stream.csv
python tools/kernel_breakdown.py $O/kernel_stats_${tag}_single_stream.csv 13 > $O/kernel_breakdown_$tag.txt 2>&1
python tools/kernel_breakdown.py $O/kernel_stats_$tag.csv 13 >> $O/kernel_breakdown_$tag.txt 2>&1
cat $O/kernel_breakdown_$tag.txt
