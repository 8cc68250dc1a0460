#!/bin/bash
# SQ / LDS counters of the kernels whose name contains <match>, under any command (counters only, three passes).
# bash tools/pmc_cmd.sh <tag> <match> <python script + args, relative to the repo root>   -> gpurun_out/<tag>_pmc.txt
TAG=$1; MATCH=$2; shift; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_a /tmp/pc_b /tmp/pc_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pc_a -- python $SCRIPT "$@" > /tmp/pc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace -d /tmp/pc_b -- python $SCRIPT "$@" > /tmp/pc_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d /tmp/pc_c -- python $SCRIPT "$@" > /tmp/pc_c.log 2>&1
python $ROOT/tools/rocpd_pmc.py --match "$MATCH" $(find /tmp/pc_a /tmp/pc_b /tmp/pc_c -name "*.db") > $OUT/${TAG}_pmc.txt 2>&1
tail -3 /tmp/pc_a.log /tmp/pc_b.log /tmp/pc_c.log | cut -c1-200 >> $OUT/${TAG}_pmc.txt
cat $OUT/${TAG}_pmc.txt | cut -c1-1200
