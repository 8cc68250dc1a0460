#!/bin/bash
# The round's committed measurement set in one GPU call: bench line + rocprofv3 summaries (profile_round.sh), PMC traffic / MFMA busy (pmc_round.sh),
# ALA epoch summary and gaps (profile_ala.sh).  bash tools/profile_final.sh r04_z ; then copy gpurun_out/<tag>_* into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04_z}
bash tools/profile_round.sh $TAG 2>&1 | tail -25
bash tools/pmc_round.sh $TAG 2>&1 | tail -12
bash tools/profile_ala.sh $TAG 2>&1 | tail -45
