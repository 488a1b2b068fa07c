#!/bin/bash
# the multi-dataset workload under the one-rank RCCL process group: usage flags host to host (gloo twin group,
# the default) against the round-4 device read (PV2_GSYNC_HOST_FLAGS=0).  Output: gpurun_out/ppt_pg_*.log
mkdir -p gpurun_out
export PV2_BENCH_FORCE_DIST=1
for hf in 1 0; do
  PV2_GSYNC_HOST_FLAGS=$hf timeout 170 python bench.py --workload ppt --steps 10 --warmup 3 \
      --no-cpu-baseline --no-kernel-timing > gpurun_out/ppt_pg_host$hf.log 2>&1
  echo "host_flags=$hf exit $?"; tail -1 gpurun_out/ppt_pg_host$hf.log | cut -c1-400
done
