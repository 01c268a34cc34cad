#!/bin/bash
# PMC passes over one conv shape / a few tile configs (separate passes: SQ has 8 slots)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/bench_conv.py --only $1 --impls $2 --reps 3"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $R/gpurun_out/pmc1 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $R/gpurun_out/pmc2 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $R/gpurun_out/pmc3 --output-format csv -- $CMD > /dev/null 2>&1
cd $R
for d in pmc1 pmc2 pmc3; do python tools/pmc_summary.py gpurun_out/$d; done
