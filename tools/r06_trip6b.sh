#!/bin/bash
set -u
O=gpurun_out/r06; mkdir -p $O
b() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; }
{
echo "PG default queues, no overlap (no comm stream):    $(PV2_BENCH_FORCE_DIST=1 PV2_GSYNC_OVERLAP=0 b)"
echo "PG default queues, no wgrad side stream:           $(PV2_BENCH_FORCE_DIST=1 PV2_WGRAD_STREAM=0 b)"
echo "PG default queues, sync skipped:                   $(PV2_BENCH_FORCE_DIST=1 PV2_BENCH_SKIP_SYNC=1 b)"
echo "PG queues=3:                                       $(PV2_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=3 b)"
echo "no PG queues=2:                                    $(GPU_MAX_HW_QUEUES=2 b)"
echo "no PG queues=3:                                    $(GPU_MAX_HW_QUEUES=3 b)"
} | tee $O/one_rank_pg_b.txt
PV2_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --workload ppt 2>&1 | tail -12
