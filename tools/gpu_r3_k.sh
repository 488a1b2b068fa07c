#!/bin/bash
# Round 3, trip K: the one-launch camera inverse through the goldens (sampler bins stay bit-exact), the
# outdoor full-size test with its bound, launch map of the native path, the other two workloads' lines.
set -u
O=gpurun_out/r3k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_fused_head.py -m gpu -q --timeout 300 -k "small_inverse or full_size or golden or default" > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_subset.txt | tail -8 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-330 $O/bench_default.json; echo
timeout 300 python tools/launch_map.py > $O/launch_map.txt 2>&1; grep -v Warning $O/launch_map.txt | tail -95
timeout 300 python bench.py --workload outdoor --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_outdoor.json 2> $O/bench_outdoor.err; echo "outdoor rc=$?"; cut -c1-330 $O/bench_outdoor.json; echo
timeout 300 python bench.py --workload ppt --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_ppt.json 2> $O/bench_ppt.err; echo "ppt rc=$?"; cut -c1-330 $O/bench_ppt.json; echo
bash tools/gpu_prof.sh r3k_outdoor --workload outdoor --steps 10 --warmup 3; cp gpurun_out/prof_r3k_outdoor_kernel_stats.csv $O/kernel_stats_outdoor.csv 2>/dev/null
python tools/kernel_breakdown.py $O/kernel_stats_outdoor.csv 13
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3k/kernel_stats_outdoor.csv')))
for r in sorted(rows,key=lambda r:-int(r['TotalDurationNs']))[:28]:
    print(f"{int(r['Calls'])/13:7.1f} {int(r['TotalDurationNs'])/13/1e3:8.1f}us {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:110]}")
PY
