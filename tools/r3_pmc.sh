#!/bin/bash
# quick PMC comparison of the pipelined kernel and the two-group kernel on the default workload
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
P3="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
tools/pmc_quick.sh ${1}_pipe_1 "$P1" --no-pcie --reps 1
tools/pmc_quick.sh ${1}_pipe_2 "$P2" --no-pcie --reps 1
tools/pmc_quick.sh ${1}_pipe_3 "$P3" --no-pcie --reps 1
D2D_NO_PIPE=1 tools/pmc_quick.sh ${1}_nopipe_1 "$P1" --no-pcie --reps 1
D2D_NO_PIPE=1 tools/pmc_quick.sh ${1}_nopipe_2 "$P2" --no-pcie --reps 1
D2D_NO_PIPE=1 tools/pmc_quick.sh ${1}_nopipe_3 "$P3" --no-pcie --reps 1
