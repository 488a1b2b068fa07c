#!/bin/bash
set -u
O=gpurun_out/r06; mkdir -p $O
PV2_WGRAD_STREAM=0 PV2_BN_FINISH=1 bash tools/gpu_prof.sh r06f1 --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r06f1_kernel_stats.csv $O/kernel_stats_finish1_single.csv
PV2_WGRAD_STREAM=0 PV2_BN_FINISH=0 bash tools/gpu_prof.sh r06f0 --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r06f0_kernel_stats.csv $O/kernel_stats_finish0_single.csv
python tools/kernel_breakdown.py $O/kernel_stats_finish1_single.csv 13 > $O/kernel_breakdown_finish1.txt 2>&1
python tools/kernel_breakdown.py $O/kernel_stats_finish0_single.csv 13 > $O/kernel_breakdown_finish0.txt 2>&1
head -16 $O/kernel_breakdown_finish1.txt $O/kernel_breakdown_finish0.txt
