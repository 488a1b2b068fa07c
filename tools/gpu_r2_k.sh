#!/bin/bash
set -u
O=gpurun_out/r2k; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_half.py -q -x -k "spconv16" > $O/pytest_half.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_half.txt | cut -c1-200
timeout 300 python tools/bench_spconv16.py > $O/spconv16_variants.txt 2>&1; cat $O/spconv16_variants.txt | cut -c1-250
timeout 300 python tools/bench_spconv16.py --perm > $O/spconv16_variants_perm.txt 2>&1; cat $O/spconv16_variants_perm.txt | cut -c1-250
