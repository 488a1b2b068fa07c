#!/bin/bash
set -u
O=gpurun_out/r06; mkdir -p $O
for f in tests/test_gpu_golden.py tests/test_gpu_fused_head.py tests/test_gpu_trainer.py; do
  timeout 900 python -m pytest $f -m gpu -q -x 2>&1 | tail -5
done
for i in 1 2; do
  PV2_FUSED_RAY_LOSS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rayloss=0', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
  PV2_FUSED_RAY_LOSS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rayloss=1', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
done
