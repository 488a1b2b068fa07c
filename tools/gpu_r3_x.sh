#!/bin/bash
set -u
O=gpurun_out/r3x; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_conv_pr.py -m gpu -q --timeout 300 -k "sparse_first or first_layer or golden or full_size or native" > $O/pytest_a.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_a.txt | tail -6 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c90-330 $O/bench_default.json; echo
timeout 300 python bench.py --workload outdoor --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_outdoor.json 2> $O/bench_outdoor.err; echo "outdoor rc=$?"; cut -c90-300 $O/bench_outdoor.json; echo
