#!/bin/bash
# Trip W: the folded final convolution - kernel tests, goldens, A/B against the materialised volume.
set -u
O=gpurun_out/w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused_head.py tests/test_gpu_golden.py tests/test_gpu_sidestream.py tests/test_gpu_kernels.py -m gpu -q -k "fold or golden or full_size or side_stream or pointwise or fused_head" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.txt | tail -25
for i in a b; do
  for fold in 1 0; do
    PV2_FOLD_FINAL_CONV=$fold timeout 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 > $O/bench_fold${fold}_$i.json 2> $O/bench.err
    echo "f32 fold=$fold $i: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_fold${fold}_$i.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench_fold${fold}_$i.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_fold${fold}_$i.json)"
  done
done
tail -3 $O/bench.err
