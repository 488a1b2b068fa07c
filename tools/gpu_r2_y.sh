#!/bin/bash
set -u
O=gpurun_out/y; mkdir -p $O
timeout 150 python bench.py --workload ppt --steps 14 --warmup 7 --no-cpu-baseline --no-kernel-timing > $O/bench_ppt.json 2>$O/bench_ppt.err; echo "ppt rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_ppt.json) $(grep -o '"final_loss": [0-9.a-zN]*' $O/bench_ppt.json)"; tail -2 $O/bench_ppt.err
timeout 200 python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/y/bench_f32.json').read().strip().splitlines()[-1])
print(d["ms_per_step"], d["host_enqueue_ms_per_step"])
print("roofline:", {k:d["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_us","launches")})
for k in d["kernels"][:6]: print({a:(round(k[a],4) if isinstance(k[a],float) else k[a]) for a in ("kernel","launches","avg_us","tflops","frac_of_mfma_peak")})
PY
PV2_FOLD_FINAL_CONV=0 timeout 150 python tools/probe_dense_unet.py > $O/probe_dense_unet.txt 2>&1; echo "probe rc=$?"; head -40 $O/probe_dense_unet.txt | cut -c1-170
grep -A34 "Self CUDA" $O/probe_dense_unet.txt | cut -c1-60,120-330 | head -36
