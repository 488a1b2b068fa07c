#!/bin/bash
set -u
O=gpurun_out/r3f; mkdir -p $O
bash tools/gpu_pmc_micro.sh r3f_busy "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" $GRAFT_REPO_ROOT/tools/micro_conv_pr.py
bash tools/gpu_pmc_micro.sh r3f_stall "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" $GRAFT_REPO_ROOT/tools/micro_conv_pr.py
for t in busy stall; do cp gpurun_out/pmc_r3f_${t}_by_kernel.csv $O/ 2>/dev/null; grep -E "kernel,|spconv_fwd_lds|row_reduce|wgrad" gpurun_out/pmc_r3f_${t}_by_kernel.csv | cut -c1-60,120-400; done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --kernel-table $O/kernel_table.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3f/bench_default.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step","final_loss")})
    print("roofline:", {k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_us","launches","traffic")})
    for k in d["kernels"][:6]: print({a:(round(k[a],4) if isinstance(k[a],float) else k[a]) for a in ("kernel","launches","avg_us","tflops","frac_of_mfma_peak") if a in k})
except Exception as e: print("bench parse failed", e); print(open('gpurun_out/r3f/bench_default.err').read()[-1500:])
PY
head -16 $O/kernel_table.txt
