#!/bin/bash
set -u
O=gpurun_out/r06; mkdir -p $O
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
for v in 0 1; do
  PV2_NATIVE_UNET16=$v timeout 300 python bench.py --amp bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('amp bf16 bs2 native16=$v', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fp32 bs2', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
for v in 0 1; do
  PV2_NATIVE_UNET16=$v timeout 400 python bench.py --amp bf16 --scenes-per-gpu 8 --views 5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('amp bf16 shipped native16=$v', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
done
timeout 400 python bench.py --scenes-per-gpu 8 --views 5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fp32 shipped', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
