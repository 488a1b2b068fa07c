#!/bin/bash
# Round 3, trip Q: fused losses and ray set-up - parity, goldens, bench, launch map.
set -u
O=gpurun_out/r3q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ray_setup.py tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_sidestream.py -m gpu -q --timeout 300 -k "ray_setup or surface_losses or full_size or golden or sidestream or side_stream" > $O/pytest_a.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/pytest_a.txt | tail -12 | cut -c1-500
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c90-330 $O/bench_default.json; echo; tail -2 $O/bench_default.err
timeout 300 python tools/launch_map.py > $O/launch_map.txt 2>&1; grep -v Warning $O/launch_map.txt | grep -E "launches  host|total" | head -30
