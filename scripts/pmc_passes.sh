#!/bin/bash
# PMC passes over the f16 scan at the bench shape; one counter set per rocprofv3 run.
#   scripts/pmc_passes.sh OUTDIR [queries]     (THR_DENSE_F16=p exported beforehand selects round 1's kernel)
set -e
D=$1; Q=${2:-1536}
mkdir -p $D
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $D -o $1 -- python3 scripts/pmc_scan.py run f16 $Q > $D/$1.log 2>&1; }
run sqA "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run sqB "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES"
run sqC "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
run sqD "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_SMEM"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run fetch "FETCH_SIZE"
