#!/bin/bash
set -u
b() { timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2), d.get('launches_per_step'))"; }
python -m pytest tests/test_gpu_fused_head.py tests/test_gpu_ray_epilogue.py tests/test_gpu_golden.py tests/test_gpu_conv_pr.py tests/test_gpu_trainer.py tests/test_gpu_narrow_head.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -5
for i in 1 2 3; do echo "bench: $(b)"; done
