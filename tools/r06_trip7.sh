#!/bin/bash
set -u
b() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2), d.get('backward_side_stream'))"; }
for c in 0 192 128 96 64 32; do echo "PV2_WGRAD_CUS=$c: $(PV2_WGRAD_CUS=$c b)"; done
echo "PV2_WGRAD_PRIORITY=1: $(PV2_WGRAD_PRIORITY=1 b)"
echo "side stream off: $(PV2_WGRAD_STREAM=0 b)"
