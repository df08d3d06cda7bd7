#!/bin/bash
# counters of the stochastic vector kernels (run on the GPU box from the repository root): bash tools/st_pmc.sh > gpurun_out/st_pmc.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st_pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/st_pmc/p$i -- python $R/tools/st_probe_big.py > /tmp/st_pmc_o$i.txt 2>&1
done
python $R/tools/pmc_summary.py /tmp/st_pmc _st
python $R/tools/pmc_summary.py /tmp/st_pmc k_v4_quantize
python $R/tools/pmc_summary.py /tmp/st_pmc k_v4_scale
