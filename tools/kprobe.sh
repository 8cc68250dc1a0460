#!/bin/bash
# SQ / LDS counters of one conv layer under the forward-kernel configurations given to tools/kprobe.py (counters only).
# bash tools/kprobe.sh <tag> <kprobe.py args...>   -> gpurun_out/<tag>_kprobe.txt
TAG=$1; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp_a /tmp/kp_b /tmp/kp_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/kp_a -- python $ROOT/tools/kprobe.py "$@" > /tmp/kp_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace -d /tmp/kp_b -- python $ROOT/tools/kprobe.py "$@" > /tmp/kp_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d /tmp/kp_c -- python $ROOT/tools/kprobe.py "$@" > /tmp/kp_c.log 2>&1
python $ROOT/tools/rocpd_pmc.py --match conv_ $(find /tmp/kp_a /tmp/kp_b /tmp/kp_c -name "*.db") > $OUT/${TAG}_kprobe.txt 2>&1
tail -3 /tmp/kp_a.log /tmp/kp_b.log /tmp/kp_c.log | cut -c1-200 >> $OUT/${TAG}_kprobe.txt
cat $OUT/${TAG}_kprobe.txt | cut -c1-900
