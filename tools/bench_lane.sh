#!/bin/bash
# the low-rank configurations on the pair kernel (3) and the per-lane kernel (2), three batch sizes: tools/bench_lane.sh <outfile>
out=${1:-gpurun_out/bench_lane.txt}
: > $out
run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-solver --steps 5 --warmup 2 "$@" 2>>$out.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); f = r['roofline']
        print(f\"{r['config']['workload'][:28]:28s} {f['kernel']:40s} F={r['config']['fibers_per_dim_per_gpu']:8d} {f['avg_launch_ms']:8.4f} ms/launch  {r['value']:.3e} nodes/s  frac {f['frac']:.3f}\")
" >> $out; }
for W in dubins3d lqg2d rossler3d; do
  for F in 16384 131072 1048576; do
    run --workload $W --fibers $F --variant 3
    run --workload $W --fibers $F --variant 2
  done
done
cat $out
