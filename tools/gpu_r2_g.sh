#!/bin/bash
# Round-2 trip G: whole GPU suite with the strict bounds, smoke, bench fp32 + --amp bf16.
set -u
O=gpurun_out/r2g; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -s > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.txt | tail -3
grep -E "^\{'loss|^\{'out|bin flips" $O/pytest.txt | cut -c1-900
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"; cut -c1-260 $O/bench_f32.json; echo
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp bf16 > $O/bench_amp_bf16.json 2> $O/bench_amp.err; echo "amp rc=$?"; cut -c1-260 $O/bench_amp_bf16.json; echo; tail -3 $O/bench_amp.err
