#!/bin/bash
# configs[3] profile set (taken on the final commit, hipGraph ON): the bench line, the per-layer table of the
# instrumented eager iterations, and the rocprofv3 kernel summary of the captured run.  bash tools/profile_c4.sh r04_c4
TAG=${1:-r04_c4}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
python bench.py --workload c4 --steps 40 --warmup 10 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
FEDICRA_BENCH_TABLE=$OUT/${TAG}_per_layer_roofline.txt python bench.py --workload c4 --roofline-only > $OUT/${TAG}_roofline.json 2>> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c4
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -- python $ROOT/bench.py --workload c4 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline \
    > /tmp/prof_c4.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_c4 -name "*.db" | head -1) > $OUT/${TAG}_bench_kernel_stats.csv
tail -c 1200 $OUT/${TAG}_bench.json; echo; head -8 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-160
