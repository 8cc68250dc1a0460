#!/bin/bash
# MFMA utilisation and LDS bank conflicts per kernel (counters only; separate passes).  bash tools/pmc_util.sh r01_f
TAG=${1:-r01_f}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --clients-per-gpu 1"
rm -rf /tmp/pu_a /tmp/pu_b
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pu_a -- $CMD > /tmp/pu_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/pu_b -- $CMD > /tmp/pu_b.log 2>&1
python $ROOT/tools/rocpd_pmc.py $(find /tmp/pu_a -name "*.db" | head -1) $(find /tmp/pu_b -name "*.db" | head -1) > $OUT/${TAG}_pmc_util.txt 2>&1
head -40 $OUT/${TAG}_pmc_util.txt | cut -c1-250; tail -3 /tmp/pu_a.log | cut -c1-200
