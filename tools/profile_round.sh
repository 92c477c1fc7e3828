set -e
# TAG names the outputs (r01_g: the default bench batch, 2^20 fibers per launch; run with TAG=r01_f BENCH_ARGS="--fibers 131072"
# C3SC_PMC_NODES=5373952 for the 2^17 batch of the earlier profiles)
TAG=${TAG:-r01_g}
BENCH_ARGS=${BENCH_ARGS:-}
export C3SC_PMC_NODES=${C3SC_PMC_NODES:-42991616}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 400 python -m pytest tests -x -q -m gpu > $O/f_gpu_tests.txt 2>&1; tail -2 $O/f_gpu_tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/f_smoke.txt 2>&1; tail -1 $O/f_smoke.txt
timeout -k 10 400 python bench.py $BENCH_ARGS --steps 10 --warmup 2 > $O/f_bench.json 2> $O/f_bench.err; echo bench done
rm -rf $O/prof_r1f $O/pmc_f1 $O/pmc_f2 $O/pmc_f3 $O/pmc_f4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1f -- python3 bench.py $BENCH_ARGS --steps 5 --warmup 1 --no-cpu-baseline > $O/f_bench_rocprof.json 2> $O/f_rocprof.err; echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f1 -- python3 bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/f_pmc1.err; echo pmc1 done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_f2 -- python3 bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/f_pmc2.err; echo pmc2 done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_f3 -- python3 bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/f_pmc3.err; echo pmc3 done
timeout -k 10 300 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $O/pmc_f4 -- python3 bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/f_pmc4.err; echo pmc4 done
python tools/make_pmc_json.py $O/${TAG}_fiber_pair_pmc.json $O/pmc_f1 $O/pmc_f2 $O/pmc_f3 $O/pmc_f4
find $O/prof_r1f -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
# keep the merged output small
find $O/prof_r1f $O/pmc_f1 $O/pmc_f2 $O/pmc_f3 $O/pmc_f4 -name "*.csv" -size +2M -delete
