#!/bin/bash
set -u
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv_pr.py -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'])
        for k in d['kernels']:
            if 'os_kernel' in k['kernel'] or '<1, 1, 4>' in k['kernel']: print('   ', k['kernel'][:60], k['avg_us'])
"
