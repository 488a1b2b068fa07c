#!/bin/bash
set -u
O=gpurun_out/r06; mkdir -p $O
b() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],2), round(d.get('host_enqueue_ms_per_step'),2))"; }
timeout 600 python -m pytest tests/test_gpu_grad_overlap.py -m gpu -q 2>&1 | grep -E "passed|failed"
{
echo "ppt, no process group:                    $(b --workload ppt)"
echo "ppt, one-rank group (overlapped slabs):   $(PV2_BENCH_FORCE_DIST=1 b --workload ppt)"
echo "ppt, one-rank group, no overlap:          $(PV2_BENCH_FORCE_DIST=1 PV2_GSYNC_OVERLAP=0 b --workload ppt)"
echo "ppt, no process group:                    $(b --workload ppt)"
echo "ppt, one-rank group (overlapped slabs):   $(PV2_BENCH_FORCE_DIST=1 b --workload ppt)"
} | tee -a $O/one_rank_pg_c.txt
