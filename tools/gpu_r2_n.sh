#!/bin/bash
set -u
O=gpurun_out/r2n; mkdir -p $O
timeout 300 python tools/profile_host.py --amp bf16 > $O/host_amp_noprefetch.txt 2>&1; grep -E "10 steps|backbone_fwd" $O/host_amp_noprefetch.txt | tail -2; sed -n '/ncalls/,$p' $O/host_amp_noprefetch.txt | head -14 | cut -c1-150
timeout 300 python tools/profile_host.py --amp bf16 --prefetch > $O/host_amp_prefetch.txt 2>&1; grep -E "10 steps|backbone_fwd" $O/host_amp_prefetch.txt | tail -2; sed -n '/ncalls/,$p' $O/host_amp_prefetch.txt | head -22 | cut -c1-150
