#!/bin/bash
# Round-end measurement trip: default bench (with live kernel timing + CPU baseline), its rocprofv3
# kernel stats, smoke(), and the outdoor workload.
set -u
mkdir -p gpurun_out
timeout 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"
bash tools/gpu_prof.sh final --steps 10 --warmup 3
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 120 python bench.py --workload outdoor --steps 10 --warmup 3 > gpurun_out/final_bench_outdoor.json 2>/dev/null; echo "outdoor rc=$?"
cut -c1-300 gpurun_out/final_bench.json; echo; cut -c80-260 gpurun_out/final_bench_outdoor.json
