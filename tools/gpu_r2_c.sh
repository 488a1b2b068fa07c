#!/bin/bash
# Round-2 trip C: output-stationary conv - parity tests, per-shape A/B, bench A/B.
set -u
O=gpurun_out/r2i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 300 python tools/bench_spconv_os.py > $O/spconv_os_ab.txt 2>&1; echo "ab rc=$?"; cat $O/spconv_os_ab.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_os.json 2> $O/bench_os.err; echo "bench os rc=$?"
PV2_SPCONV_OSL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_scatter.json 2> $O/bench_scatter.err; echo "bench scatter rc=$?"
cut -c1-330 $O/bench_os.json; echo; cut -c1-330 $O/bench_scatter.json; echo
