#!/bin/bash
# Round-2 trip H: the 16-bit sparse kernels - their tests, then the --amp bf16 bench line.
set -u
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_half.py -q -x > $O/pytest_half.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_half.txt | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp bf16 > $O/bench_amp_bf16.json 2> $O/bench_amp.err; echo "amp rc=$?"; cut -c1-400 $O/bench_amp_bf16.json; echo; tail -5 $O/bench_amp.err
