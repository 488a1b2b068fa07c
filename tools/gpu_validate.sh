#!/bin/bash
# Round 3, trip T: full suite + default bench + kernel stats (two streams / one) at the current commit.
set -u
O=gpurun_out/validate; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -12 | cut -c1-300
timeout 500 python bench.py --steps 20 --warmup 5 --kernel-table $O/kernel_table.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-360 $O/bench_default.json; echo
bash tools/gpu_prof.sh val --steps 10 --warmup 3; cp gpurun_out/prof_val_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh val_single --steps 10 --warmup 3; cp gpurun_out/prof_val_single_kernel_stats.csv $O/kernel_stats_single_stream.csv 2>/dev/null
python tools/kernel_breakdown.py $O/kernel_stats_single_stream.csv 13
python - <<'PY'
import json
d=json.loads(open('gpurun_out/val/bench_default.json').read().strip().splitlines()[-1])
print({k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_us","launches","traffic","alg_bytes_per_launch")})
print(d.get("cpu_baseline",{}).get("value"), d.get("optimizer"), d.get("sparse_backbone"))
PY
PV2_WGRAD_STREAM=0 bash tools/gpu_pmc.sh val_outdoor_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload outdoor --steps 6 --warmup 2
cp gpurun_out/pmc_val_outdoor_mfma_by_kernel.csv $O/pmc_outdoor_mfma_by_kernel.csv 2>/dev/null; head -8 $O/pmc_outdoor_mfma_by_kernel.csv | cut -c1-200
