cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
for dbg in 0 8192; do
rm -rf $O/pw
C3SC_DBG=$dbg timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 bench.py --workload quad10d --no-cpu-baseline --no-solver --steps 2 --warmup 1 > /dev/null 2> $O/pw.err
python - <<PY
import csv, glob
v=[float(r["Counter_Value"]) for fn in glob.glob("$O/pw/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(fn)) if "k_fiber_quad" in r["Kernel_Name"]]
print("dbg $dbg WRITE_SIZE mean MB per launch", sum(v)/len(v)*1024/1e6, "n", len(v))
PY
C3SC_DBG=$dbg python bench.py --workload quad10d --no-cpu-baseline --no-solver --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('dbg $dbg', r['roofline']['avg_launch_ms'], 'ms')
"
done
find $O/pw -name "*.csv" -size +1M -delete
