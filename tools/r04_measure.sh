#!/bin/bash
# Round 4 measurement trip A: full GPU suite, smoke(), default bench (kernel timing + CPU baseline),
# rocprofv3 kernel stats (two streams / one), kernel table, breakdown.  Outputs: gpurun_out/r04/
set -u
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -8 | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 500 python bench.py --kernel-table $O/kernel_table.txt > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"; cut -c1-420 $O/bench_f32.json; echo
bash tools/gpu_prof.sh r04 --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r04_kernel_stats.csv $O/kernel_stats_f32.csv
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh r04s --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r04s_kernel_stats.csv $O/kernel_stats_f32_single_stream.csv
python tools/kernel_breakdown.py $O/kernel_stats_f32_single_stream.csv 13 > $O/kernel_breakdown.txt 2>&1
python tools/kernel_breakdown.py $O/kernel_stats_f32.csv 13 >> $O/kernel_breakdown.txt 2>&1
head -16 $O/kernel_breakdown.txt
