#!/bin/bash
# Round profile: the driver's bench line + rocprofv3 kernel summaries of (a) the same command and (b) the instrumented
# iterations alone (bench.py --roofline-only: the launch mix the roofline object prices).  Run on the GPU box from the
# repo root: bash tools/profile_round.sh r02_a ; results land in gpurun_out/<tag>_*
TAG=${1:-r02_a}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
FEDICRA_BENCH_TABLE=$OUT/${TAG}_per_layer_roofline.txt FEDICRA_BENCH_VERBOSE=1 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k /tmp/prof_r
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dice --no-3d \
    > /tmp/prof_k.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_k -name "*.db" | head -1) > $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_r -- python $ROOT/bench.py --roofline-only > /tmp/prof_r.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_r -name "*.db" | head -1) > $OUT/${TAG}_roofline_kernel_stats.csv
tail -c 1500 $OUT/${TAG}_bench.json; echo; tail -c 300 $OUT/${TAG}_bench.err; head -12 $OUT/${TAG}_roofline_kernel_stats.csv | cut -c1-200
