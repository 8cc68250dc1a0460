#!/bin/bash
# SQ / LDS counters of the autocorrelation kernels (counters only; separate passes).  bash tools/xcprobe.sh <tag>
TAG=${1:-xc}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/xp_a /tmp/xp_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/xp_a -- python $ROOT/tools/xcbench.py --no-direct > /tmp/xp_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace -d /tmp/xp_b -- python $ROOT/tools/xcbench.py --no-direct > /tmp/xp_b.log 2>&1
python $ROOT/tools/rocpd_pmc.py --match xcorr_partial $(find /tmp/xp_a /tmp/xp_b -name "*.db") > $OUT/${TAG}_xcprobe.txt 2>&1
tail -2 /tmp/xp_a.log /tmp/xp_b.log | cut -c1-200 >> $OUT/${TAG}_xcprobe.txt
cat $OUT/${TAG}_xcprobe.txt | cut -c1-1200
