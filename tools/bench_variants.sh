#!/bin/bash
# bench the kernel variants side by side: tools/bench_variants.sh <outfile>
out=${1:-gpurun_out/bench_variants.txt}
: > $out
run() { python bench.py --no-cpu-baseline --no-solver --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); f = r['roofline']
        print(f\"{r['config']['workload'][:40]:40s} {f['kernel']:44s} F={r['config']['fibers_per_dim_per_gpu']:8d} {f['avg_launch_ms']:8.3f} ms/launch  {r['value']:.3e} nodes/s  frac {f['frac']:.3f}\")
" >> $out; }
run --workload car7d --fibers 131072 --variant 3
run --workload car7d --fibers 131072 --variant 4
run --workload car7d --variant 3
run --workload car7d --variant 4
run --workload quad10d --variant 0
run --workload quad10d --variant 4
run --workload scar4d --variant 0
run --workload scar4d --variant 4
run --workload lqg6d --variant 3
run --workload lqg6d --variant 4
cat $out
