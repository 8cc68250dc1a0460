#!/bin/bash
# Idle-gap analysis of the timed training rounds (captured iterations + aggregation rounds, bf16): rocprofv3 kernel + memory-copy trace of
# the driver's command without its side legs, then tools/rocpd_gaps.py over the steady state.  bash tools/profile_gaps.sh <tag>
TAG=${1:-r04_z}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_g -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --windows 2 --no-cpu-baseline --no-dice --no-roofline --no-fp32 --no-resident --no-3d > /tmp/prof_g.log 2>&1
DB=$(find /tmp/prof_g -name "*.db" | head -1)
( tail -c 300 /tmp/prof_g.log; echo; python $ROOT/tools/rocpd_gaps.py $DB --tail-ms 175 --top 12; python $ROOT/tools/rocpd_timeline.py $DB --tail-ms 100 --min-gap-us 60 --context 2 ) > $OUT/${TAG}_train_gaps.txt 2>&1
cut -c1-200 $OUT/${TAG}_train_gaps.txt | tail -150
