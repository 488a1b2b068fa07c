#!/bin/bash
# the GPU suite file by file + as a whole -> gpurun_out/r06/pytest_gpu.txt (the record copied to profiles/)
O=gpurun_out/r06; mkdir -p $O
: > $O/pytest_gpu.txt
for f in tests/test_gpu_*.py; do
  case $f in *zero_edit*) continue;; esac
  echo "== $f" >> $O/pytest_gpu.txt
  timeout 420 python -m pytest $f -m gpu -q 2>&1 | grep -vE "Warning|warnings.warn|^  |^$|pin_memory|Docs:" >> $O/pytest_gpu.txt; echo "$f rc=${PIPESTATUS[0]}"
done
echo "== whole suite" >> $O/pytest_gpu.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -1 | tee -a $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/pytest_gpu.txt
