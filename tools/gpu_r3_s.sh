#!/bin/bash
set -u
O=gpurun_out/r3s; mkdir -p $O
for args in "--amp bf16" "--raw-points" "--workload ppt" "--workload outdoor"; do
  tag=$(echo $args | tr -d ' -')
  timeout 300 python bench.py $args --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$args rc=$?"; cut -c90-300 $O/bench_$tag.json; echo; tail -1 $O/bench_$tag.err | cut -c1-200
done
