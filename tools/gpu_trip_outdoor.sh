#!/bin/bash
# One GPU trip for the outdoor (nuScenes-shaped) path: parity tests, MIOpen solver search for the
# SimpleConv3D shapes (results copied back so they can ship in miopen_cache/), then the bench.
set -u
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_golden.py -q -m gpu -k "outdoor" -x 2>&1 | tail -15 > gpurun_out/outdoor_tests.log
echo "tests rc=$?" >> gpurun_out/outdoor_tests.log
PV2_MIOPEN_SEARCH=1 timeout 400 python bench.py --workload outdoor --steps 3 --warmup 2 --no-kernel-timing > gpurun_out/outdoor_search.log 2>&1
echo "search rc=$?" >> gpurun_out/outdoor_search.log
rm -rf gpurun_out/miopen_cache_new && cp -r miopen_cache gpurun_out/miopen_cache_new
timeout 300 python bench.py --workload outdoor --steps 10 --warmup 3 --kernel-table gpurun_out/outdoor_kernel_table.txt > gpurun_out/outdoor_bench.json 2> gpurun_out/outdoor_bench.err
echo "bench rc=$?" >> gpurun_out/outdoor_bench.err
timeout 200 python bench.py --workload outdoor --steps 10 --warmup 3 --no-graph --no-kernel-timing > gpurun_out/outdoor_bench_nograph.json 2>> gpurun_out/outdoor_bench.err
tail -3 gpurun_out/outdoor_tests.log; tail -2 gpurun_out/outdoor_search.log; cat gpurun_out/outdoor_bench.json | cut -c1-600
