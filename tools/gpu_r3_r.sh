#!/bin/bash
set -u
O=gpurun_out/r3r; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ray_setup.py -m gpu -q --timeout 200 > $O/pytest_a.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/pytest_a.txt | tail -8 | cut -c1-400
timeout 300 python tools/launch_map.py > $O/launch_map.txt 2>&1; grep -v Warning $O/launch_map.txt | sed -n 1,140p | cut -c1-120
