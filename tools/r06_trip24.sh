#!/bin/bash
set -u
b() { timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2))"; }
echo "== split below 512"; PV2_DCONV_KSPLIT_WGS=512 python tools/bench_dense_conv.py 2>&1 | tail -11
for i in 1 2 3; do
echo "256: $(b)"
echo "512: $(PV2_DCONV_KSPLIT_WGS=512 b)"
done
