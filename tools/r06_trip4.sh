#!/bin/bash
set -u
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_head.py -m gpu -q 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/bench_rows1.json
python - <<PY
import json
d=json.load(open("$O/bench_rows1.json"))
print(d["ms_per_step"], d.get("host_enqueue_ms_per_step"))
for k in d.get("kernels", []):
    if "field" in k.get("kernel","") or "fold" in k.get("kernel",""):
        print("   ", k["kernel"][:50], k["avg_us"], k["tflops"])
PY
