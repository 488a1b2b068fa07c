#!/bin/bash
set -u
O=gpurun_out/r3w; mkdir -p $O
timeout 300 python tools/launch_map.py > $O/launch_map.txt 2>&1; grep -A30 "copy / gather" $O/launch_map.txt | cut -c1-250
