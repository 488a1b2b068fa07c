#!/bin/bash
# Round 3, trip O: native executor with input-feature gradients, sync-free block masking, outdoor host profile + bench.
set -u
O=gpurun_out/r3o; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_conv_pr.py tests/test_gpu_golden.py -m gpu -q --timeout 200 -k "native or outdoor" > $O/pytest_a.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error" $O/pytest_a.txt | tail -8 | cut -c1-400
timeout 300 python tools/profile_host.py --outdoor --prefetch > $O/host_profile_outdoor.txt 2>&1; grep -v Warning $O/host_profile_outdoor.txt | head -30 | cut -c1-170
timeout 300 python bench.py --workload outdoor --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_outdoor.json 2> $O/bench_outdoor.err; echo "outdoor rc=$?"; cut -c90-300 $O/bench_outdoor.json; echo
bash tools/gpu_prof.sh r3o_outdoor --workload outdoor --steps 10 --warmup 3; cp gpurun_out/prof_r3o_outdoor_kernel_stats.csv $O/kernel_stats_outdoor.csv 2>/dev/null
python tools/kernel_breakdown.py $O/kernel_stats_outdoor.csv 13
