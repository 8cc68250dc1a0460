#!/bin/bash
# Launch-geometry knobs of the one-tile conv kernels under the bench workload (configs[2]); bash tools/knob_sweep.sh <tag>
TAG=${1:-r02}; OUT=gpurun_out/${TAG}_knob_sweep.txt; : > $OUT
run() { echo "== $*" >> $OUT; env "$@" python bench.py --no-cpu-baseline --no-fp32 --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config']['ms_per_aggregation_round'])" >> $OUT; }
run FI_NOP=1
run FI_TARGET_BLOCKS=512
run FI_TARGET_BLOCKS=4096
run FI_MIN_BLOCKS=2048
run FI_FWD_LDS_CAP=65536
run FI_V2=0
run FI_WGRAD_BLOCKS=1024
cat $OUT
