#!/bin/bash
# Whole value-iteration sweeps of the solver under the kernel trace (timeline with the idle gaps) and with the driver's own breakdown:
#   tools/profile_sweep.sh [tag=r04] [workload=car7d] [sweeps=14]
set -e
T=${1:-r04}; W=${2:-car7d}; S=${3:-14}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/p_sw
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_sw -o sweep -- python3 $GRAFT_REPO_ROOT/tools/vi_sweep_quick.py $W $S > $O/${T}_vi_sweep_${W}_under_rocprof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/sweep_timeline.py $(find $O/p_sw -name "*.db" | head -1) > $O/${T}_vi_sweep_${W}_timeline.txt
C3SC_PROFILE=1 python tools/vi_sweep_quick.py $W $S > $O/${T}_vi_sweep_${W}.txt 2>&1
echo sweep profiles done
