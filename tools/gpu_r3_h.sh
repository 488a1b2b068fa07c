#!/bin/bash
set -u
O=gpurun_out/r3h; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_conv_pr.py tests/test_gpu_sidestream.py -m gpu -q -s --timeout 200 -k "native or side_stream" > $O/pytest_native.txt 2>&1; echo "native tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|native vs modular|Error|error" $O/pytest_native.txt | tail -12 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -q -s --timeout 300 -k "not ppt_full" > $O/pytest_golden.txt 2>&1; echo "golden rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_golden.txt | tail -8 | cut -c1-300
grep -o "float64 gradient record {[^}]*}" $O/pytest_golden.txt | cut -c1-500
for mode in "PV2_X=0" "PV2_NATIVE_UNET=0"; do
  env $mode timeout 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/bench_ab.json 2> $O/bench_ab.err
  echo "$mode: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ab.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench_ab.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_ab.json)"; tail -3 $O/bench_ab.err | cut -c1-300
done
timeout 200 python bench.py --raw-points --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/bench_raw.json 2> $O/bench_raw.err; echo "raw points: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_raw.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench_raw.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_raw.json)"; tail -2 $O/bench_raw.err | cut -c1-300
timeout 200 python tools/profile_host.py --prefetch > $O/host_profile.txt 2>&1; head -8 $O/host_profile.txt | cut -c1-200
