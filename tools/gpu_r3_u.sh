#!/bin/bash
set -u
O=gpurun_out/r3u; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_pr.py -m gpu -q --timeout 200 -k "wgrad or weight or native or unit" > $O/pytest_a.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_a.txt | tail -6 | cut -c1-300
for v in 0 1; do
  PV2_WGRAD_SMALL_TILE=$v timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --kernel-table $O/kernel_table_$v.txt > $O/bench_$v.json 2> $O/bench_$v.err; echo "small=$v rc=$?"; cut -c90-260 $O/bench_$v.json; echo
  grep -E "wgrad" $O/kernel_table_$v.txt | head -14
done
