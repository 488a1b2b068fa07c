#!/bin/bash
# Round 3, trip P: the whole -m gpu suite, smoke(), the default bench line with cpu baseline.
set -u
O=gpurun_out/r3p; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -12 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 500 python bench.py --steps 20 --warmup 5 --kernel-table $O/kernel_table.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-360 $O/bench_default.json; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p/bench_default.json').read().strip().splitlines()[-1])
print({k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_us","launches","traffic","alg_bytes_per_launch")})
print(d.get("cpu_baseline",{}).get("value"), d.get("optimizer"))
PY
