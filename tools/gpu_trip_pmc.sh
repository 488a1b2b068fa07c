#!/bin/bash
mkdir -p gpurun_out
bash tools/gpu_pmc_micro.sh dc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $GRAFT_REPO_ROOT/tools/micro_dense_one.py
bash tools/gpu_pmc_micro.sh dc_wait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $GRAFT_REPO_ROOT/tools/micro_dense_one.py
bash tools/gpu_pmc_micro.sh dc_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" $GRAFT_REPO_ROOT/tools/micro_dense_one.py
cat gpurun_out/pmc_dc_*_by_kernel.csv | grep -v "^kernel" | grep "dconv" | cut -c1-60,140-400

