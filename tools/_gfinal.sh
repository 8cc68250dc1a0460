cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04_z}
bash tools/profile_round.sh $TAG 2>&1 | tail -25
bash tools/pmc_round.sh $TAG 2>&1 | tail -12
bash tools/profile_ala.sh $TAG 2>&1 | tail -45
