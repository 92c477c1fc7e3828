#!/bin/bash
# rocprofv3 summaries of ONE bench configuration:  tools/profile_config.sh <tag> <nodes_per_launch> <fibers> <bench args...>
#   un-profiled bench line, kernel-trace stats, then separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ f64 / MFMA work, SQ busy / wait
#   shares).  Every profiled run passes --no-overlap-probe: the untimed "other launch mode" pass of bench.py spreads the d launches
#   of a step over three streams, and its overlapping dispatches would otherwise be averaged into the per-kernel rows (round 3's
#   kernel_stats.csv did not reproduce roofline.avg_launch_ms for that reason).  The program stands directly behind `--`.
set -e
TAG=$1; NODES=$2; FIB=$3; shift 3
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
A="$@ --no-cpu-baseline --no-solver --no-overlap-probe"
timeout -k 10 300 python bench.py $@ --steps 10 --warmup 2 --no-solver --no-overlap-probe > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo bench done
rm -rf $O/p_st $O/p_1 $O/p_2 $O/p_3 $O/p_4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_st -- python3 bench.py $A --steps 5 --warmup 1 > $O/${TAG}_bench_under_rocprof.json 2> $O/p_st.err; echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_1 -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/p_1.err; echo pmc1 done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_2 -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/p_2.err; echo pmc2 done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $O/p_3 -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/p_3.err; echo pmc3 done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/p_4 -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/p_4.err; echo pmc4 done
C3SC_PMC_NODES=$NODES C3SC_PMC_FIBERS=$FIB python tools/make_pmc_json.py $O/${TAG}_pmc.json $O/p_1 $O/p_2 $O/p_3 $O/p_4
find $O/p_st -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
# the check the profiles are judged by: every k_fiber_* row's AverageNs against roofline.avg_launch_ms of the same (profiled) run
python tools/check_kernel_stats.py $O/${TAG}_kernel_stats.csv $O/${TAG}_bench_under_rocprof.json | tee $O/${TAG}_kernel_stats_check.txt
find $O/p_st $O/p_1 $O/p_2 $O/p_3 $O/p_4 -name "*.csv" -size +1M -delete
