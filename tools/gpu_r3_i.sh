#!/bin/bash
set -u
O=gpurun_out/r3i; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_conv_pr.py tests/test_gpu_kernels.py tests/test_gpu_sidestream.py -m gpu -q --timeout 200 > $O/pytest_conv.txt 2>&1; echo "conv tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_conv.txt | tail -8 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --kernel-table $O/kernel_table.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3i/bench_default.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step","final_loss")})
    print("roofline:", {k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_us","launches","traffic")})
    for k in d["kernels"][:6]: print({a:(round(k[a],4) if isinstance(k[a],float) else k[a]) for a in ("kernel","launches","avg_us","tflops","frac_of_mfma_peak") if a in k})
except Exception as e: print("bench parse failed", e); print(open('gpurun_out/r3i/bench_default.err').read()[-1500:])
PY
grep -E "row_reduce|kernel " $O/kernel_table.txt | head -14
