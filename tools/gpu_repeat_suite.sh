#!/bin/bash
# The -m gpu suite (or the files given after N) N times WITHOUT pytest's output capture, so that messages
# of the HIP runtime / C++ libraries survive an abort: a check for rare failures.
# usage: tools/gpu_repeat_suite.sh [N] [pytest targets...]
set -u
N=${1:-2}; shift || true
T=${@:-tests}
O=gpurun_out/repeat; mkdir -p $O
for i in $(seq 1 $N); do
  timeout 900 python -m pytest $T -m gpu -q --timeout 400 -s -p no:faulthandler > $O/run_$i.txt 2> $O/run_$i.err
  rc=$?; echo "run $i rc=$rc"; grep -E "passed|failed" $O/run_$i.txt | tail -1 | cut -c1-200
  if [ $rc -ne 0 ]; then grep -v "amdgpu.ids" $O/run_$i.err | tail -15 | cut -c1-400; tail -5 $O/run_$i.txt | cut -c1-300; fi
done
