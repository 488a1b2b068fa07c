#!/bin/bash
python -m pytest tests/test_gpu_dense_conv.py tests/test_gpu_dense_unet.py tests/test_gpu_golden.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -5
b() { timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2))"; }
python tools/bench_dense_conv.py 2>&1 | tail -11 | tee gpurun_out/r06/dense_conv_microbench.txt
for i in 1 2 3; do echo "bench: $(b)"; done
