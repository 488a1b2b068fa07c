#!/bin/bash
set -u
O=gpurun_out/r2o; mkdir -p $O
python tools/find_syncs.py --amp --prefetch 2>&1 | tail -12 | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_kernels.py -q -x -k "indoor or sparse_first or prefetch or spconv_forward_backward or wgrad" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt | cut -c1-300
for mode in "" "--amp bf16"; do
for pf in "" "--no-prefetch"; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $mode $pf > $O/bench.json 2> $O/bench.err; echo "[$mode $pf] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench.json) $(grep -o '"final_loss": [0-9.e-]*' $O/bench.json)"
done; done
