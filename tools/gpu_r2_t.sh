#!/bin/bash
set -u
bash tools/gpu_pmc_micro.sh stall "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" $GRAFT_REPO_ROOT/tools/micro_conv_layers.py
bash tools/gpu_pmc_micro.sh lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" $GRAFT_REPO_ROOT/tools/micro_conv_layers.py
bash tools/gpu_pmc_micro.sh l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" $GRAFT_REPO_ROOT/tools/micro_conv_layers.py
for t in stall lds l2; do echo "== $t"; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/pmc_${t}_by_kernel.csv')))
for r in rows:
    if any(k in r['kernel'] for k in ('os16','wgrad16','spconv_fwd_lds','spconv_wgrad_lds')):
        print(r['kernel'][:48], {k.replace('_per_dispatch',''):v for k,v in r.items() if k!='kernel'})
PY
done
