#!/bin/bash
set -u
b() { timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2))"; }
for i in 1 2 3; do
echo "late2: $(PV2_WGRAD_LATE=2 b)"
echo "late : $(PV2_WGRAD_LATE=1 b)"
done
