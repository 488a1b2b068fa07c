#!/bin/bash
# Round-2 trip A: fused ray-march kernels stage check, GPU tests of the head, quick bench A/B.
set -u
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
timeout 300 python tools/check_fused_head.py > $O/check_fused.txt 2>&1; echo "check rc=$?"
timeout 600 python -m pytest tests/test_gpu_fused_head.py tests/test_gpu_golden.py -q -m gpu -x > $O/pytest_head.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_head.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench fused rc=$?"
PV2_FUSED_HEAD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_modular.json 2> $O/bench_modular.err; echo "bench modular rc=$?"
cut -c1-400 $O/bench_fused.json; echo; cut -c1-400 $O/bench_modular.json; echo
grep -E "err|rel" $O/check_fused.txt | awk '{ if ($NF+0 > 1e-4) print }' | head -40
