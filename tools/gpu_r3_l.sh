#!/bin/bash
# Round 3, trip L: the narrow-decoder render head (csrc/raymarch_narrow.hip) against its oracle, the
# refactored coarse-pass code of the ScanNet head, outdoor goldens through the fused route, outdoor bench.
set -u
O=gpurun_out/r3l; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_narrow_head.py -m gpu -q -x --timeout 120 -s > $O/pytest_narrow.txt 2>&1; echo "narrow rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^[0-9]+ \[|Error|error" $O/pytest_narrow.txt | tail -14 | cut -c1-1500
timeout 600 python -m pytest tests/test_gpu_fused_head.py tests/test_gpu_golden.py -m gpu -q --timeout 300 > $O/pytest_heads.txt 2>&1; echo "heads rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_heads.txt | tail -8 | cut -c1-300
timeout 300 python bench.py --workload outdoor --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_outdoor.json 2> $O/bench_outdoor.err; echo "outdoor rc=$?"; cut -c1-330 $O/bench_outdoor.json; echo; tail -3 $O/bench_outdoor.err
PV2_NARROW_HEAD=0 timeout 300 python bench.py --workload outdoor --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_outdoor_modular.json 2> $O/bench_outdoor_modular.err; echo "outdoor modular rc=$?"; cut -c1-330 $O/bench_outdoor_modular.json; echo
bash tools/gpu_prof.sh r3l_outdoor --workload outdoor --steps 10 --warmup 3; cp gpurun_out/prof_r3l_outdoor_kernel_stats.csv $O/kernel_stats_outdoor.csv 2>/dev/null
python tools/kernel_breakdown.py $O/kernel_stats_outdoor.csv 13
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3l/kernel_stats_outdoor.csv')))
for r in sorted(rows,key=lambda r:-int(r['TotalDurationNs']))[:16]:
    print(f"{int(r['Calls'])/13:7.1f} {int(r['TotalDurationNs'])/13/1e3:8.1f}us {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:110]}")
PY
