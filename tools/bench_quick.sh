#!/bin/bash
# one line per run: tools/bench_quick.sh <workload> [bench args]
W=$1; shift
python bench.py --workload $W --no-cpu-baseline --no-solver --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); ro = r['roofline']; print('$W', '%.3e' % r['value'], round(ro['avg_launch_ms'], 4), round(ro['frac'], 3), ro.get('kernel'))
"
