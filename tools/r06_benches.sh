#!/bin/bash
# Round 6: the other workloads on the final tree -> gpurun_out/r06/workloads.txt + bench_*.json
set -u
O=gpurun_out/r06; mkdir -p $O
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | grep "^{" > $O/bench_$name.json; python - <<PY
import json
d=json.load(open("$O/bench_$name.json"))
print("%-34s %7.2f ms/step  %8.1f scenes/s  host %6.2f  %s"%("$name", d["ms_per_step"], d["value"], d.get("host_enqueue_ms_per_step",0), d.get("dtype")))
PY
}
{
run default_20 --steps 20 --warmup 5
run outdoor --workload outdoor --steps 20 --warmup 5
run ppt --workload ppt --steps 20 --warmup 5
run rawpoints --raw-points --steps 20 --warmup 5
run amp_bf16_bs2 --amp bf16 --steps 20 --warmup 5
run amp_fp16_bs2 --amp fp16 --steps 20 --warmup 5
run shipped_f32 --scenes-per-gpu 8 --views 5 --steps 10 --warmup 3
run shipped_bf16 --amp bf16 --scenes-per-gpu 8 --views 5 --steps 10 --warmup 3
PV2_BENCH_FORCE_DIST=1 run one_rank_pg --steps 20 --warmup 5
} | tee $O/workloads.txt
