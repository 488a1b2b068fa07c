#!/bin/bash
# Round-2 trip D: BN without atomics, host-side tile prefixes, autocast-immune kernels.
set -u
O=gpurun_out/r2k; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench.json; echo
bash tools/gpu_prof.sh r2d --steps 10 --warmup 3; cp gpurun_out/prof_r2d_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
timeout 200 python tools/profile_host.py > $O/host_profile.txt 2>&1; grep "backbone_fwd" $O/host_profile.txt | tail -2
