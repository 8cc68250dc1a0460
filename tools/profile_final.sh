#!/bin/bash
# The round's committed measurement set in one GPU call: bench line + rocprofv3 summaries (profile_round.sh), PMC traffic / MFMA busy
# (pmc_round.sh), ALA epoch summary and gaps (profile_ala.sh), idle gaps of the timed rounds (profile_gaps.sh), the 3D workloads
# (profile_c4.sh for configs[3]; the bench line of configs[4]).  bash tools/profile_final.sh r05_z ; then copy gpurun_out/<tag>_* into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r05_z}
# PMC passes first: the bench line that follows then finds a traffic file taken on ITS sources (roofline.traffic is refused otherwise)
bash tools/pmc_round.sh $TAG 2>&1 | tail -12
cd $GRAFT_REPO_ROOT
cp gpurun_out/${TAG}_pmc_traffic.json profiles/
bash tools/profile_round.sh $TAG 2>&1 | tail -25
bash tools/profile_ala.sh $TAG 2>&1 | tail -45
bash tools/profile_gaps.sh $TAG 2>&1 | tail -40
bash tools/profile_c4.sh ${TAG}_c4 2>&1 | tail -12
cd $GRAFT_REPO_ROOT
python bench.py --workload c5 --steps 40 --warmup 12 --no-cpu-baseline > gpurun_out/${TAG}_c5_bench.json 2> gpurun_out/${TAG}_c5_bench.err
tail -c 600 gpurun_out/${TAG}_c5_bench.json
