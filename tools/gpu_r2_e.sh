#!/bin/bash
# Round-2 trip E: full-size parity tests, bench with cpu_baseline, PMC passes (traffic + MFMA).
set -u
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py -q -m gpu -x -s -k "full_size" > $O/pytest_full.txt 2>&1; echo "pytest rc=$?"; grep -E "bin flips|passed|failed|Error" $O/pytest_full.txt | cut -c1-1500
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json; echo
bash tools/gpu_pmc.sh r2e_fetch "FETCH_SIZE" --steps 5 --warmup 2
bash tools/gpu_pmc.sh r2e_write "WRITE_SIZE" --steps 5 --warmup 2
bash tools/gpu_pmc.sh r2e_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" --steps 5 --warmup 2
cp gpurun_out/pmc_r2e_*_by_kernel.csv $O/ 2>/dev/null
python tools/pmc_to_json.py gpurun_out/pmc_r2e_fetch_by_kernel.csv gpurun_out/pmc_r2e_write_by_kernel.csv "$(cat .commit_id 2>/dev/null || echo unknown)" $O/pmc_fetch_write_per_kernel.json
tail -3 gpurun_out/pmc_r2e_mfma.log
