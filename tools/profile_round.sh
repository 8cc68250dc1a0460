#!/bin/bash
# Round profile: bench lines (co-located default, single client, fp32) + rocprofv3 kernel summaries.  Run on the GPU box
# from the repo root: bash tools/profile_round.sh r01_e ; results land in gpurun_out/<tag>_*
TAG=${1:-r01_e}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench_bf16.json 2> $OUT/${TAG}_bench_bf16.err
python bench.py --clients-per-gpu 1 > $OUT/${TAG}_bench_bf16_1client.json 2>> $OUT/${TAG}_bench_bf16.err
python bench.py --dtype fp32 > $OUT/${TAG}_bench_fp32.json 2>> $OUT/${TAG}_bench_bf16.err
cd /tmp && export TMPDIR=/tmp
for c in 2 1; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -- python $ROOT/bench.py --clients-per-gpu $c --no-cpu-baseline \
      > /tmp/prof_$c.log 2>&1
  DB=$(find /tmp/prof_$c -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py $DB > $OUT/${TAG}_bench_bf16_${c}client_kernel_stats.csv
done
tail -c 600 $OUT/${TAG}_bench_bf16.json; echo; tail -c 300 $OUT/${TAG}_bench_bf16.err
