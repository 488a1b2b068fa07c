#!/bin/bash
# Round 3, trip A: first light of the product-row conv path, the deterministic weight gradient and
# the fused conv + BatchNorm units - parity tests, per-layer A/B, step-level A/B, rocprofv3 stats.
set -u
O=gpurun_out/r3a; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_conv_pr.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 120 > $O/pytest_conv.txt 2>&1; echo "conv tests rc=$?"; tail -25 $O/pytest_conv.txt | cut -c1-300
timeout 200 python tools/bench_spconv32.py > $O/spconv_ab.txt 2>&1; echo "ab rc=$?"; cat $O/spconv_ab.txt | cut -c1-260
timeout 600 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_gpu_conv_pr.py --deselect tests/test_gpu_kernels.py > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -12 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --kernel-table $O/kernel_table.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3a/bench_default.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step","final_loss")})
    print("roofline:", {k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_us","launches","traffic")})
    for k in d["kernels"][:10]: print({a:(round(k[a],4) if isinstance(k[a],float) else k[a]) for a in ("kernel","launches","avg_us","tflops","frac_of_mfma_peak") if a in k})
except Exception as e: print("bench parse failed", e); print(open('gpurun_out/r3a/bench_default.err').read()[-1500:])
PY
for mode in "PV2_CONVBN=0" "PV2_CONV_PR=0 PV2_WGRAD_DET=0 PV2_SPCONV_OS=0" "PV2_CONV_PR=subm"; do
  env $mode timeout 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/bench_ab.json 2> $O/bench_ab.err
  echo "$mode: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ab.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench_ab.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_ab.json)"
done
bash tools/gpu_prof.sh r3a --steps 10 --warmup 3; cp gpurun_out/prof_r3a_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
python tools/kstats.py $O/kernel_stats.csv 13 45 2>/dev/null
