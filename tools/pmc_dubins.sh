cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/pmc_d1 $O/pmc_d2
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d1 -- python3 bench.py --workload dubins3d --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/d_pmc1.err; echo pmc1 done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_d2 -- python3 bench.py --workload dubins3d --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/d_pmc2.err; echo pmc2 done
python tools/make_pmc_json.py $O/dubins_pmc.json $O/pmc_d1 $O/pmc_d2
find $O/pmc_d1 $O/pmc_d2 -name "*.csv" -size +2M -delete
