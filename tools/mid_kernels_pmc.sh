#!/bin/bash
# counters of the streaming kernels between 0.64 and 0.75 of the HBM peak (run on the GPU box from the repository root):
#   bash tools/mid_kernels_pmc.sh > gpurun_out/r05_mid_kernels_pmc.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/mk_pmc
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/mk_pmc/p$i -- python $R/tools/mid_kernels_probe.py > /tmp/mk_pmc_o$i.txt 2>&1 < /dev/null
done
for k in k_m4_transpose k_m4_quantize_strip k_m4_mvm8 k_v8_quantize k_v4_quantize; do timeout 120 python $R/tools/pmc_summary.py /tmp/mk_pmc $k < /dev/null; done
