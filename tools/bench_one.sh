#!/bin/bash
# one configuration, one line: tools/bench_one.sh <workload> <fibers> [variant]
python3 bench.py --workload $1 --fibers $2 --variant ${3:-0} --steps 5 --warmup 2 --no-cpu-baseline --no-solver 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); f = r['roofline']
        print('%-44s F=%8d  %.4f ms/launch  %.3e nodes/s  frac %.3f' % (f['kernel'], r['config']['fibers_per_dim_per_gpu'], f['avg_launch_ms'], r['value'], f['frac']))
"
