#!/bin/bash
# Round-2 closing record (v10): border-class constant part of the dense first layer on top of v9.
set -u
O=gpurun_out/z; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -8
timeout 400 python bench.py --kernel-table $O/kernel_table_f32.txt > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"; cut -c1-330 $O/bench_f32.json; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/z/bench_f32.json').read().strip().splitlines()[-1])
print("roofline:", {k:d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_us","launches","traffic")})
for k in d["kernels"][:8]: print({a:(round(k[a],4) if isinstance(k[a],float) else k[a]) for a in ("kernel","launches","avg_us","tflops","frac_of_mfma_peak")})
print("cpu_baseline:", d.get("cpu_baseline",{}).get("value"))
PY
bash tools/gpu_prof.sh z_f32 --steps 10 --warmup 3; cp gpurun_out/prof_z_f32_kernel_stats.csv $O/kernel_stats_f32.csv 2>/dev/null
PV2_WGRAD_STREAM=0 bash tools/gpu_prof.sh z_f32_single --steps 10 --warmup 3; cp gpurun_out/prof_z_f32_single_kernel_stats.csv $O/kernel_stats_f32_single_stream.csv 2>/dev/null
timeout 150 python bench.py --amp bf16 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 5 > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_bf16.json)"
