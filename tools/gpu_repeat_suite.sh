#!/bin/bash
# The whole -m gpu suite N times with the HIP runtime's error log on (AMD_LOG_LEVEL=1 prints only errors):
# a check for rare failures.  usage: tools/gpu_repeat_suite.sh [N]
set -u
N=${1:-2}
O=gpurun_out/repeat; mkdir -p $O
for i in $(seq 1 $N); do
  AMD_LOG_LEVEL=1 timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/run_$i.txt 2> $O/run_$i.err
  echo "run $i rc=$?"; tail -1 $O/run_$i.txt | cut -c1-200; grep -v "^$" $O/run_$i.err | grep -iv "amdgpu.ids" | head -12 | cut -c1-300
done
