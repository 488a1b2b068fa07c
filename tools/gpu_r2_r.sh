#!/bin/bash
set -u
O=gpurun_out/r2r; mkdir -p $O
for w in outdoor ppt; do for mode in "" "--amp bf16"; do
timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $mode > $O/bench_${w}.json 2> $O/bench_${w}.err; echo "[$w $mode] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${w}.json) $(grep -o '"value": [0-9.]*' $O/bench_${w}.json) $(grep -o '"final_loss": [0-9.e-]*' $O/bench_${w}.json) $(grep -o '"loss_sane": [a-z]*' $O/bench_${w}.json)"; tail -2 $O/bench_${w}.err | cut -c1-200
done; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --amp fp16 > $O/bench_fp16.json 2> $O/bench_fp16.err; echo "[indoor fp16] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_fp16.json) $(grep -o '"final_loss": [0-9.e-]*' $O/bench_fp16.json) $(grep -o '"loss_sane": [a-z]*' $O/bench_fp16.json)"; tail -2 $O/bench_fp16.err | cut -c1-200
