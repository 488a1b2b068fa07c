#!/bin/bash
# Round 5 measurement trip: the other workloads and operating points on the final tree (no kernel timing pass: the
# per-kernel numbers are in profiles/r05_kernel_table.txt), plus the route of DESIGN 3.2c end to end.
set -u
O=gpurun_out/r05; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", round(d["ms_per_step"],2), round(d["host_enqueue_ms_per_step"],2), round(d["value"],1), d["unit"], "loss_sane" , d.get("loss_sane"))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run default_20 --steps 20 --warmup 5
run outdoor --workload outdoor --steps 10 --warmup 3
run ppt --workload ppt --steps 10 --warmup 3
run rawpoints --raw-points --steps 20 --warmup 5
run amp_bf16_bs2 --amp bf16 --steps 20 --warmup 5
run shipped_f32 --scenes-per-gpu 8 --views 5 --steps 6 --warmup 2
run shipped_bf16 --scenes-per-gpu 8 --views 5 --amp bf16 --steps 6 --warmup 2
PV2_CONV_OSM=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PV2_CONV_OSM=1 (every eligible conv output-stationary)', round(d['ms_per_step'],2), 'loss_sane', d.get('loss_sane'))"
