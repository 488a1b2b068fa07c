#!/bin/bash
set -u
O=gpurun_out/r3g; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -12 | cut -c1-300
grep -E "float64 gradient record|loss error per run" $O/pytest_gpu.txt | cut -c1-400
timeout 200 python -m pytest tests/test_gpu_golden.py -m gpu -q -s -k "full_size" 2>&1 | grep -E "float64|loss error|bin flips" | cut -c1-600 | tail -12
for mode in "PV2_X=0" "PV2_CL_MAXPOOL=0"; do
  env $mode timeout 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/bench_ab.json 2> $O/bench_ab.err
  echo "$mode: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ab.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench_ab.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_ab.json)"
done
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh r3g_single --steps 10 --warmup 3; cp gpurun_out/prof_r3g_single_kernel_stats.csv $O/kernel_stats_single_stream.csv 2>/dev/null
python tools/kernel_breakdown.py $O/kernel_stats_single_stream.csv 13
python tools/kstats.py $O/kernel_stats_single_stream.csv 13 30 2>/dev/null | cut -c1-150
