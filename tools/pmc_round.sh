#!/bin/bash
# HBM traffic per kernel family for the round (two PMC passes; counters only, no other trace domains) over the launch mix
# of the roofline object (bench.py --roofline-only).  Run on the GPU box from the repo root: bash tools/pmc_round.sh r02_a
TAG=${1:-r02_a}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--roofline-only"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -- python $ROOT/bench.py $ARGS > /tmp/pmc_$c.log 2>&1
done
rm -rf /tmp/pmc_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d /tmp/pmc_MFMA -- python $ROOT/bench.py $ARGS > /tmp/pmc_MFMA.log 2>&1
python $ROOT/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) \
    "$ARGS" "${KEY:-c3-unet_lc-12x3x512-bf16}" $(find /tmp/pmc_MFMA -name "*.db" | head -1) /tmp/pmc_FETCH_SIZE.log > $OUT/${TAG}_pmc_traffic.json
head -c 1500 $OUT/${TAG}_pmc_traffic.json
