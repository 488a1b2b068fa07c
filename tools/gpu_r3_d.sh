#!/bin/bash
set -u
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -12 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench.json)"
