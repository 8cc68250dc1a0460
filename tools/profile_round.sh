#!/bin/bash
# Round profile: the driver's bench line + a rocprofv3 kernel summary of the same command.  Run on the GPU box from the
# repo root: bash tools/profile_round.sh r02_a ; results land in gpurun_out/<tag>_*
TAG=${1:-r02_a}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
FEDICRA_BENCH_VERBOSE=1 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline \
    --no-fp32 > /tmp/prof_k.log 2>&1
DB=$(find /tmp/prof_k -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $DB > $OUT/${TAG}_bench_kernel_stats.csv
tail -c 1500 $OUT/${TAG}_bench.json; echo; tail -c 300 $OUT/${TAG}_bench.err; head -30 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-220
