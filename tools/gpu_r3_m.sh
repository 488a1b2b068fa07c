#!/bin/bash
# Round 3, trip M: the narrow head in its MFMA formulation - parity, outdoor goldens, outdoor bench, profile.
set -u
O=gpurun_out/r3m; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_narrow_head.py -m gpu -q -x --timeout 120 -s > $O/pytest_narrow.txt 2>&1; echo "narrow rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^[0-9]+ \[|Error|error" $O/pytest_narrow.txt | tail -8 | cut -c1-900
timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -q --timeout 300 -k "outdoor" > $O/pytest_outdoor.txt 2>&1; echo "outdoor goldens rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_outdoor.txt | tail -8 | cut -c1-300
for v in 2; do PV2_NARROW_BWD=$v timeout 200 python bench.py --workload outdoor --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 2>/dev/null | cut -c90-260; PV2_NARROW_BWD=$v bash tools/gpu_prof.sh r3m_v$v --workload outdoor --steps 6 --warmup 2; grep -E "narrow_field_bwd|narrow_field_fwd" gpurun_out/prof_r3m_v${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-; done
timeout 300 python bench.py --workload outdoor --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_outdoor.json 2> $O/bench_outdoor.err; echo "outdoor rc=$?"; cut -c1-330 $O/bench_outdoor.json; echo; tail -3 $O/bench_outdoor.err
bash tools/gpu_prof.sh r3m_outdoor --workload outdoor --steps 10 --warmup 3; cp gpurun_out/prof_r3m_outdoor_kernel_stats.csv $O/kernel_stats_outdoor.csv 2>/dev/null
python tools/kernel_breakdown.py $O/kernel_stats_outdoor.csv 13
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3m/kernel_stats_outdoor.csv')))
for r in sorted(rows,key=lambda r:-int(r['TotalDurationNs']))[:12]:
    print(f"{int(r['Calls'])/13:7.1f} {int(r['TotalDurationNs'])/13/1e3:8.1f}us {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:110]}")
PY
