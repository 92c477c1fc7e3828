#!/bin/bash
# Round-4 profiles: the three configurations the verdict names (kernel stats that reproduce the bench line + PMC passes), then
# whole solver sweeps (kernel trace -> timeline).   tools/profile_r04.sh [tag=r04]
set -e
T=${1:-r04}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
bash tools/profile_config.sh ${T}_car7d 42991616 1048576 --workload car7d
bash tools/profile_config.sh ${T}_quad10d 3276800 131072 --workload quad10d
bash tools/profile_config.sh ${T}_scar4d 5242880 131072 --workload scar4d
echo profiles done
