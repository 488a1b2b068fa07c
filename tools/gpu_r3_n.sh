#!/bin/bash
# Round 3, trip N: lean optimizer steps, outdoor host profile, outdoor / indoor / ppt bench lines.
set -u
O=gpurun_out/r3n; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_narrow_head.py -m gpu -q --timeout 120 -k "lean or narrow or small_inverse" > $O/pytest_a.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error" $O/pytest_a.txt | tail -8 | cut -c1-400
timeout 300 python tools/profile_host.py --outdoor --prefetch > $O/host_profile_outdoor.txt 2>&1; grep -v Warning $O/host_profile_outdoor.txt | head -48 | cut -c1-170
for w in outdoor; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; cut -c90-300 $O/bench_$w.json; echo; done
grep -o "\"optimizer\": \"[^\"]*\"" $O/bench_outdoor.json
