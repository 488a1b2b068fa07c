#!/bin/bash
# The counters behind tools/r06_fake_split.sh: vector instructions and MFMA-busy cycles of the product-row
# kernels with the split in place and with the split free.
set -u
for lib in real fake; do
  [ $lib = fake ] && export PV2_PROBE_LIB=libponderv2_fake.so
  bash tools/gpu_pmc_micro.sh r06split_${lib}_valu "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES" $GRAFT_REPO_ROOT/tools/micro_conv_pr.py
  bash tools/gpu_pmc_micro.sh r06split_${lib}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" $GRAFT_REPO_ROOT/tools/micro_conv_pr.py
done
for f in gpurun_out/pmc_r06split_*_by_kernel.csv; do echo "== $f"; grep -E "Kernel|spconv_fwd_lds|spconv_wgrad_split|row_reduce" $f | cut -c1-260; done
