#!/bin/bash
# Round 6, trip 1: BatchNorm backward sums in the grad-input row reduce + wider combine: parity tests, then A/B
set -u
O=gpurun_out/r06; mkdir -p $O
for f in tests/test_gpu_conv_pr.py tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_dense_unet.py; do
  timeout 600 python -m pytest $f -m gpu -q -x 2>&1 | tail -3
done
for i in 1 2; do
  PV2_BN_BWD_FUSED=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused=0', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
  PV2_BN_BWD_FUSED=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused=1', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
done
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh r06a --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r06a_kernel_stats.csv $O/kernel_stats_single_a.csv
python tools/kernel_breakdown.py $O/kernel_stats_single_a.csv 13 | head -16
