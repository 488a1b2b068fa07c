#!/bin/bash
# Round-2 trip I: profile of the --amp bf16 step (kernel stats).
set -u
O=gpurun_out/r2i; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_half.py -q -x -k spunet -s > $O/pytest_half.txt 2>&1; echo "pytest rc=$?"; grep -E "cosine|passed|failed" $O/pytest_half.txt | cut -c1-300
bash tools/gpu_prof.sh amp16 --steps 10 --warmup 3 --amp bf16
cp gpurun_out/prof_amp16_kernel_stats.csv $O/kernel_stats_amp.csv; rm -rf gpurun_out/prof_amp16
head -45 $O/kernel_stats_amp.csv | cut -c1-220
