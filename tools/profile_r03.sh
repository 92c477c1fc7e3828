#!/bin/bash
# Round-3 profiles: the headline launch (kernel stats + PMC, via tools/profile_r02.sh) and whole solver sweeps (kernel trace)
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
bash tools/profile_r02.sh r03_car7d 42991616 1048576 --workload car7d
rm -rf $O/p_sw
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_sw -o sweep -- python3 $GRAFT_REPO_ROOT/tools/vi_sweep_quick.py car7d 14 > $O/r03_vi_sweep_car7d_under_rocprof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/sweep_timeline.py $(find $O/p_sw -name "*.db" | head -1) > $O/r03_vi_sweep_car7d_timeline.txt
C3SC_PROFILE=1 python tools/vi_sweep_quick.py car7d 14 > $O/r03_vi_sweep_car7d.txt 2>&1
C3SC_HOST_CROSS=1 C3SC_PROFILE=1 python tools/vi_sweep_quick.py car7d 14 > $O/r03_vi_sweep_car7d_host_driven.txt 2>&1
python bench.py > $O/r03_default_bench.json 2> $O/r03_default_bench.err
echo profiles done
