#!/bin/bash
# Alternate the full-size fixtures in one process (tools/stress_fixtures.py), then the whole suite, then the bench line.
set -u
O=gpurun_out/stress; mkdir -p $O
timeout 240 python tools/stress_fixtures.py 10 > $O/default.txt 2> $O/default.err; echo "stress rc=$? rounds=$(grep -c '^round' $O/default.txt)"; grep -i "fault\|error" $O/default.err | head -3 | cut -c1-200
timeout 600 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -5 | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 --kernel-table $O/kernel_table.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-340 $O/bench_default.json; echo
