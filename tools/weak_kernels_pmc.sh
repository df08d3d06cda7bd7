#!/bin/bash
# counters of the two streaming kernels at ~0.5 of the HBM peak (run on the GPU box from the repository root):
#   bash tools/weak_kernels_pmc.sh > gpurun_out/r04_weak_kernels_pmc.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/wk_pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" \
           "TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_BUSY TCC_REQ"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d /tmp/wk_pmc/p$i -- python $R/tools/mvmf32_probe.py > /tmp/wk_pmc_o$i.txt 2>&1 < /dev/null
done
timeout 120 python $R/tools/pmc_summary.py /tmp/wk_pmc k_m4_mvm_f32 < /dev/null
timeout 120 python $R/tools/pmc_summary.py /tmp/wk_pmc k_v4_scale_and_add_st < /dev/null
timeout 120 python $R/tools/pmc_summary.py /tmp/wk_pmc "k_v4_scale_and_add<" < /dev/null
timeout 120 python $R/tools/pmc_summary.py /tmp/wk_pmc "k_v4_scale_and_add_blk" < /dev/null
