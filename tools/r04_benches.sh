#!/bin/bash
# Round 4 measurement trip C: the other workloads and operating points, micro-benchmarks
set -u
O=gpurun_out/r04; mkdir -p $O
run() { tag=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["ms_per_step"],2), round(d["host_enqueue_ms_per_step"],2), round(d["value"],1), d["unit"])
PY
}
run default_20 --steps 20 --warmup 5
run outdoor --workload outdoor --steps 10 --warmup 3
run ppt --workload ppt --steps 10 --warmup 3
run rawpoints --raw-points --steps 20 --warmup 5
run amp_bf16_bs2 --amp bf16 --steps 20 --warmup 5
run shipped_f32 --scenes-per-gpu 8 --views 5 --steps 6 --warmup 2
run shipped_bf16 --scenes-per-gpu 8 --views 5 --amp bf16 --steps 6 --warmup 2
run fp32_mfma_ab --steps 20 --warmup 5 --no-kernel-timing
PV2_FP32_MFMA=1 timeout 400 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PV2_FP32_MFMA=1 (all products on the fp32 MFMA)', round(d['ms_per_step'],2))"
timeout 300 python tools/bench_dense_conv.py 10 > $O/dense_conv_microbench.txt 2>&1; tail -3 $O/dense_conv_microbench.txt | cut -c1-120
./tools/micro/bf16_split_probe > $O/bf16_split_probe.txt 2>&1; head -4 $O/bf16_split_probe.txt
timeout 200 python tools/check_cells_node.py > $O/cells_node_check.txt 2>&1; tail -3 $O/cells_node_check.txt
