#!/bin/bash
# Kernel summary + idle-gap analysis of the captured ALA epoch (8 batches of 12 x 3 x 512^2, bf16): bash tools/profile_ala.sh <tag>
TAG=${1:-r04_ala}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_a
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_a -- python $ROOT/tools/ala_bench.py --size 512 --in-chns 3 --graph-only --rounds 12 > /tmp/prof_a.log 2>&1
DB=$(find /tmp/prof_a -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $DB > $OUT/${TAG}_ala_kernel_stats.csv
( tail -2 /tmp/prof_a.log; python $ROOT/tools/rocpd_gaps.py $DB --tail-ms 66 --top 25 ) > $OUT/${TAG}_ala_gaps.txt 2>&1
cat $OUT/${TAG}_ala_gaps.txt | cut -c1-220; head -30 $OUT/${TAG}_ala_kernel_stats.csv | cut -c1-160
