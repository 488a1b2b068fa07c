#!/bin/bash
# Round-2 final record (v9: folded final convolution on): full GPU suite, default bench line with
# live kernel timing + CPU baseline, rocprofv3 kernel stats, the two PMC passes behind
# roofline.traffic, --amp bf16 lines (fold on / off, twice each), ppt + outdoor workloads, the
# multi-process path with one rank, smoke().
set -u
O=gpurun_out/x; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -8
timeout 400 python bench.py --kernel-table $O/kernel_table_f32.txt > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"; cut -c1-330 $O/bench_f32.json; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/x/bench_f32.json').read().strip().splitlines()[-1])
print("roofline:", {k:d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_us","launches","traffic")})
for k in d["kernels"][:12]: print({a:(round(k[a],4) if isinstance(k[a],float) else k[a]) for a in ("kernel","launches","avg_us","tflops","frac_of_mfma_peak")})
print("cpu_baseline:", d.get("cpu_baseline",{}).get("value"))
PY
bash tools/gpu_prof.sh x_f32 --steps 10 --warmup 3; cp gpurun_out/prof_x_f32_kernel_stats.csv $O/kernel_stats_f32.csv 2>/dev/null
bash tools/gpu_pmc.sh x_fetch "FETCH_SIZE" --steps 6 --warmup 2 > /dev/null
bash tools/gpu_pmc.sh x_write "WRITE_SIZE" --steps 6 --warmup 2 > /dev/null
python tools/pmc_to_json.py gpurun_out/pmc_x_fetch_by_kernel.csv gpurun_out/pmc_x_write_by_kernel.csv "$(cat tools/.commit 2>/dev/null || echo unknown)" $O/pmc_fetch_write_per_kernel.json
cp gpurun_out/pmc_x_fetch_by_kernel.csv $O/pmc_FETCH_SIZE_by_kernel.csv; cp gpurun_out/pmc_x_write_by_kernel.csv $O/pmc_WRITE_SIZE_by_kernel.csv
for i in a b; do
  for fold in 1 0; do
    PV2_FOLD_FINAL_CONV=$fold timeout 200 python bench.py --amp bf16 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 5 > $O/bench_bf16_fold${fold}_$i.json 2> $O/bench_bf16.err
    echo "bf16 fold=$fold $i: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16_fold${fold}_$i.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench_bf16_fold${fold}_$i.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_bf16_fold${fold}_$i.json)"
  done
done
bash tools/gpu_prof.sh x_bf16 --amp bf16 --steps 10 --warmup 3; cp gpurun_out/prof_x_bf16_kernel_stats.csv $O/kernel_stats_amp_bf16.csv 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 150 python bench.py --workload outdoor --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_outdoor.json 2>/dev/null; echo "outdoor rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_outdoor.json)"
timeout 150 python bench.py --workload ppt --steps 14 --warmup 7 --no-cpu-baseline --no-kernel-timing > $O/bench_ppt.json 2>$O/bench_ppt.err; echo "ppt rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ppt.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_ppt.json)"
PV2_BENCH_FORCE_DIST=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist(1 rank, flat sync) rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_dist1.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_dist1.json)"
