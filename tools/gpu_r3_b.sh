#!/bin/bash
set -u
O=gpurun_out/r3b; mkdir -p $O
timeout 200 python tools/debug_modes.py run_ponder_indoor > $O/debug_indoor.txt 2>&1; cat $O/debug_indoor.txt | cut -c1-700
timeout 200 python tools/launch_map.py > $O/launch_map.txt 2>&1; cat $O/launch_map.txt | cut -c1-200
timeout 200 python tools/profile_host.py --prefetch > $O/host_profile.txt 2>&1; head -60 $O/host_profile.txt | cut -c1-200
