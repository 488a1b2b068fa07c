#!/bin/bash
set -u
O=gpurun_out/r3v; mkdir -p $O
timeout 300 python tools/profile_host.py --prefetch > $O/host_profile.txt 2>&1; grep -v Warning $O/host_profile.txt | head -75 | cut -c1-150
timeout 200 python bench.py --workload ppt --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 2>/dev/null | grep -o '"sparse_backbone": "[^"]*"\|"ms_per_step": [0-9.]*'
