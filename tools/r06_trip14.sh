#!/bin/bash
set -u
b() { timeout 400 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2))
        for k in d.get('kernels',[]):
            if 'field' in k['kernel']: print('   ', k['kernel'][:60], k['launches'], round(k['avg_us'],2))"; }
python -m pytest tests/test_gpu_fused_head.py tests/test_gpu_golden.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -5
for i in 1 2; do echo "bench: $(b)"; done
