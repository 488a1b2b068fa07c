#!/bin/bash
set -u
b() { timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2))"; }
python -m pytest tests/test_gpu_dense_unet.py tests/test_gpu_dense_conv.py -m gpu -q 2>&1 | tail -3
echo "fp32 bs2:                     $(b --steps 20 --warmup 5)"
echo "amp bf16 bs2, dense fp32:     $(PV2_DENSE_AMP=0 b --amp bf16 --steps 20 --warmup 5)"
echo "amp bf16 bs2, dense bf16:     $(b --amp bf16 --steps 20 --warmup 5)"
echo "fp32 shipped:                 $(b --scenes-per-gpu 8 --views 5 --steps 10 --warmup 3)"
echo "amp bf16 shipped, dense fp32: $(PV2_DENSE_AMP=0 b --amp bf16 --scenes-per-gpu 8 --views 5 --steps 10 --warmup 3)"
echo "amp bf16 shipped, dense bf16: $(b --amp bf16 --scenes-per-gpu 8 --views 5 --steps 10 --warmup 3)"
