#!/bin/bash
# Round 4 measurement trip B: PMC passes of the default bench (one counter group per pass)
set -u
bash tools/gpu_pmc.sh r04_fetch "FETCH_SIZE" --steps 6 --warmup 2 > /dev/null 2>&1
bash tools/gpu_pmc.sh r04_write "WRITE_SIZE" --steps 6 --warmup 2 > /dev/null 2>&1
PV2_WGRAD_STREAM=0 bash tools/gpu_pmc.sh r04_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 6 --warmup 2 > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc_r04_fetch_by_kernel.csv gpurun_out/pmc_r04_write_by_kernel.csv 64a2dc3 gpurun_out/r04_pmc_fetch_write_per_kernel.json
head -12 gpurun_out/pmc_r04_mfma_by_kernel.csv | cut -c1-220
python - <<PY
import json
d=json.load(open("gpurun_out/r04_pmc_fetch_write_per_kernel.json"))
print(d["kernel_source_hash"], len(d["kernels"]))
for k,v in list(d["kernels"].items())[:8]: print(k[:70], v)
PY
