#!/bin/bash
# Round-2 trip B: whole GPU suite, host profile of a step, rocprof kernel stats of the bench.
set -u
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.txt
timeout 300 python tools/profile_host.py > $O/host_profile.txt 2>&1; echo "host profile rc=$?"; grep "backbone_fwd" $O/host_profile.txt | tail -2
bash tools/gpu_prof.sh r2b --steps 10 --warmup 3; cp gpurun_out/prof_r2b_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
head -40 $O/kernel_stats.csv | cut -c1-150
