#!/bin/bash
set -u
O=gpurun_out/r3y; mkdir -p $O
for w in "" "--workload outdoor"; do for p in "" 1; do
  PV2_BENCH_MAIN_PRIORITY=$p timeout 300 python bench.py $w --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*' | tr '\n' ' '; echo " [$w prio=$p]"
done; done
