#!/bin/bash
# rocprofv3 counter passes over tools/gemm_probe.py (run on the GPU box from the repository root):
#   bash tools/gemm_pmc.sh > gpurun_out/gemm_pmc.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/gemm_pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/gemm_pmc/p$i -- python $R/tools/gemm_probe.py > /tmp/gemm_pmc_o$i.txt 2>&1
done
python $R/tools/pmc_summary.py /tmp/gemm_pmc k_m4_gemm
