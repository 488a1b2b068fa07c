#!/bin/bash
# Round 3 checkpoint: the whole -m gpu suite in the default selection, the default bench line with live
# kernel timing, rocprofv3 kernel stats (two streams and one), and the PMC passes (HBM-side traffic,
# MFMA pipe occupancy) of the same sources.
set -u
O=gpurun_out/checkpoint; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -12 | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_golden.py -m gpu -q -s -k "full_size" 2>&1 | grep -oE "float64 gradient record \{[^}]*\}|loss error per run.*" | cut -c1-420
timeout 400 python bench.py --steps 20 --warmup 5 --kernel-table $O/kernel_table.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-420 $O/bench_default.json; echo
bash tools/gpu_prof.sh ckpt --steps 10 --warmup 3; cp gpurun_out/prof_ckpt_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh ckpt_single --steps 10 --warmup 3; cp gpurun_out/prof_ckpt_single_kernel_stats.csv $O/kernel_stats_single_stream.csv 2>/dev/null
python tools/kernel_breakdown.py $O/kernel_stats_single_stream.csv 13
PV2_WGRAD_STREAM=0 bash tools/gpu_pmc.sh ckpt_fetch "FETCH_SIZE" --steps 6 --warmup 2
PV2_WGRAD_STREAM=0 bash tools/gpu_pmc.sh ckpt_write "WRITE_SIZE" --steps 6 --warmup 2
PV2_WGRAD_STREAM=0 bash tools/gpu_pmc.sh ckpt_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 6 --warmup 2
cp gpurun_out/pmc_ckpt_*_by_kernel.csv $O/ 2>/dev/null
python tools/pmc_to_json.py gpurun_out/pmc_ckpt_fetch_by_kernel.csv gpurun_out/pmc_ckpt_write_by_kernel.csv $(cat tools/.commit) $O/pmc_fetch_write_per_kernel.json
