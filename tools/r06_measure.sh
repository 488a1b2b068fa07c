#!/bin/bash
# Round 6 measurement trip: full GPU suite, smoke(), default bench (live kernel timing + CPU baseline),
# rocprofv3 kernel stats (two streams / one), kernel table, breakdown, launch map, PMC passes (one counter group
# per pass, --pmc with --kernel-trace only), then the bench line again with the fresh PMC file.  Outputs: gpurun_out/r06/
set -u
O=gpurun_out/r06; mkdir -p $O
: > $O/pytest_gpu.txt
for f in tests/test_gpu_*.py; do
  case $f in *zero_edit*) continue;; esac
  echo "== $f" >> $O/pytest_gpu.txt
  timeout 420 python -m pytest $f -m gpu -q >> $O/pytest_gpu.txt 2>&1; echo "$f rc=$?"
done
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | cut -c1-200
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee -a $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 600 python bench.py --kernel-table $O/kernel_table.txt > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"; cut -c1-300 $O/bench_f32.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_default_20.json 2>/dev/null; cut -c1-260 $O/bench_default_20.json; echo
bash tools/gpu_prof.sh r06 --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r06_kernel_stats.csv $O/kernel_stats_f32.csv
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh r06s --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/prof_r06s_kernel_stats.csv $O/kernel_stats_f32_single_stream.csv
python tools/kernel_breakdown.py $O/kernel_stats_f32_single_stream.csv 13 > $O/kernel_breakdown.txt 2>&1
python tools/kernel_breakdown.py $O/kernel_stats_f32.csv 13 >> $O/kernel_breakdown.txt 2>&1
head -16 $O/kernel_breakdown.txt
timeout 300 python tools/launch_map.py > $O/launch_map.txt 2>&1; grep -n "total launches" $O/launch_map.txt
bash tools/gpu_pmc.sh r06_fetch "FETCH_SIZE" --steps 6 --warmup 2 > /dev/null 2>&1
bash tools/gpu_pmc.sh r06_write "WRITE_SIZE" --steps 6 --warmup 2 > /dev/null 2>&1
PV2_WGRAD_STREAM=0 bash tools/gpu_pmc.sh r06_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 6 --warmup 2 > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc_r06_fetch_by_kernel.csv gpurun_out/pmc_r06_write_by_kernel.csv ${PV2_COMMIT:-r06} $O/pmc_fetch_write_per_kernel.json
cp gpurun_out/pmc_r06_*_by_kernel.csv $O/ 2>/dev/null
head -6 gpurun_out/pmc_r06_mfma_by_kernel.csv | cut -c1-220
cp $O/pmc_fetch_write_per_kernel.json profiles/r06_pmc_fetch_write_per_kernel.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_f32_with_traffic.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$O/bench_f32_with_traffic.json")); print(d["ms_per_step"], d["roofline"])
PY
