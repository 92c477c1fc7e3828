cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/pmc_ic
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES --output-format csv -d $O/pmc_ic -- python3 bench.py --fibers 131072 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/ic.err; echo done
C3SC_PMC_NODES=5373952 python tools/make_pmc_json.py $O/icache_pmc.json $O/pmc_ic
find $O/pmc_ic -name "*.csv" -size +2M -delete
