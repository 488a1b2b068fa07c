#!/bin/bash
# Round-2 trip Q: the record for v7 (no host stalls; 16-bit sparse mode): bench lines, kernel stats, PMC.
set -u
O=gpurun_out/r2q; mkdir -p $O
timeout 600 python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; echo "f32 rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_f32.json)"
timeout 300 python bench.py --no-cpu-baseline --amp bf16 > $O/bench_amp_bf16.json 2> $O/bench_amp.err; echo "amp rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_amp_bf16.json)"
bash tools/gpu_prof.sh f32v7 --steps 10 --warmup 3 > /dev/null; cp gpurun_out/prof_f32v7_kernel_stats.csv $O/kernel_stats_f32.csv; rm -rf gpurun_out/prof_f32v7
bash tools/gpu_prof.sh ampv7 --steps 10 --warmup 3 --amp bf16 > /dev/null; cp gpurun_out/prof_ampv7_kernel_stats.csv $O/kernel_stats_amp.csv; rm -rf gpurun_out/prof_ampv7
bash tools/gpu_pmc.sh amp_fetch "FETCH_SIZE" --steps 4 --warmup 2 --amp bf16 > /dev/null
bash tools/gpu_pmc.sh amp_write "WRITE_SIZE" --steps 4 --warmup 2 --amp bf16 > /dev/null
bash tools/gpu_pmc.sh amp_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 4 --warmup 2 --amp bf16 > /dev/null
cp gpurun_out/pmc_amp_fetch_by_kernel.csv gpurun_out/pmc_amp_write_by_kernel.csv gpurun_out/pmc_amp_mfma_by_kernel.csv $O/ 2>/dev/null
rm -rf gpurun_out/pmc_amp_fetch gpurun_out/pmc_amp_write gpurun_out/pmc_amp_mfma
ls -la $O
