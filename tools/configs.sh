#!/bin/bash
# All SURVEY 8d configurations at the two roofline batches (2^17 and 2^20 fibers per launch) plus the SQ counters of the
# low-rank ones: tools/configs.sh <tag>.  Writes gpurun_out/<tag>_configs.txt and gpurun_out/<tag>_<workload>_sq.txt.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r04}
: > $O/${TAG}_configs.txt
for W in lqg2d dubins3d lqg6d car7d quad10d scar4d rossler3d skid5d cothrust6d perch7d; do
  for F in 131072 1048576; do
    echo "== $W fibers $F" >> $O/${TAG}_configs.txt
    timeout -k 10 240 python3 bench.py --workload $W --fibers $F --steps 5 --warmup 2 --no-cpu-baseline --no-solver --no-overlap-probe >> $O/${TAG}_configs.txt 2>> $O/${TAG}_configs.err || exit 1
  done
done
python3 - <<PY
import json
rows = []
name = None
for line in open("$O/${TAG}_configs.txt"):
    if line.startswith("=="):
        name = line.split()[1:]
    elif line.startswith("{"):
        j = json.loads(line)
        r = j["roofline"]
        rows.append((name[0], int(name[2]), j["value"], r["avg_launch_ms"], r["frac"]))
with open("$O/${TAG}_configs_table.txt", "w") as f:
    for r in rows:
        print("%-10s F=%8d  %.3e nodes/s  launch %.4f ms  frac %.3f" % r, file=f)
print(open("$O/${TAG}_configs_table.txt").read())
PY
