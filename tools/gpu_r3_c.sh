#!/bin/bash
set -u
O=gpurun_out/r3c; mkdir -p $O
timeout 400 python tools/debug_modes.py run_ponder_indoor > $O/debug_indoor.txt 2>&1; cat $O/debug_indoor.txt | cut -c1-500
timeout 200 python -m pytest tests/test_gpu_golden.py -m gpu -q -x -k "ponder_indoor_gpu or ppt_gpu" 2>&1 | tail -3 | cut -c1-300
