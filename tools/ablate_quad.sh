#!/bin/bash
# timing ablations of the fiber-quad kernel (results are wrong with a switch on; only the time is of interest)
W=${1:-quad10d}
for dbg in 0 256 512 768 1024 2048 4096 6144 6912; do
  C3SC_DBG=$dbg python bench.py --workload $W --fibers 131072 --variant 4 --steps 3 --warmup 1 --no-cpu-baseline --no-solver 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$W dbg=$dbg', '%.3f ms/launch' % r['roofline']['avg_launch_ms'])
"
done
