#!/bin/bash
set -u
O=gpurun_out/r2p; mkdir -p $O
for i in 1 2; do
for kt in "" "--no-kernel-timing"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --amp bf16 $kt > $O/bench_$i.json 2> $O/bench.err; echo "[amp $kt] rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.json) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $O/bench_$i.json)"
done; done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2p/bench_1.json'))
print(json.dumps(d.get("roofline"))[:600])
for k in d.get("kernels",[])[:8]: print({a:k[a] for a in ("kernel","launches","avg_us","tflops","frac_of_mfma_peak","frac_of_hbm_peak")})
PY
