#!/bin/bash
mkdir -p gpurun_out
export MIOPEN_USER_DB_PATH=$PWD/miopen_cache/db MIOPEN_CUSTOM_CACHE_DIR=$PWD/miopen_cache/cache
timeout 600 python -m pytest tests/test_gpu_dense_conv.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/dense_tests.txt
timeout 300 python tools/bench_dense_conv.py 10 2>&1 | tail -14 | tee gpurun_out/dense_bench.txt
PV2_DCONV_TPW=1 timeout 300 python tools/bench_dense_conv.py 10 2>&1 | tail -12 | tee gpurun_out/dense_bench_tpw1.txt
