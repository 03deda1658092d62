#!/bin/bash
TAG=$1
bash tools/pmc_pass.sh ${TAG}_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
bash tools/pmc_pass.sh ${TAG}_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
bash tools/pmc_pass.sh ${TAG}_mem FETCH_SIZE
bash tools/pmc_pass.sh ${TAG}_memw WRITE_SIZE
mkdir -p gpurun_out/$TAG
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_sq1 gpurun_out/pmc_${TAG}_sq2 gpurun_out/pmc_${TAG}_mem gpurun_out/pmc_${TAG}_memw > gpurun_out/$TAG/pmc_summary.txt 2>&1
