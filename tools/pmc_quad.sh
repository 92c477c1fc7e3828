#!/bin/bash
# SQ counters of the fiber-quad kernel: tools/pmc_quad.sh <workload> <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
W=${1:-quad10d}; TAG=${2:-r02_quad}; V=${3:-4}
rm -rf $O/pq1 $O/pq2 $O/pq3
A="--workload $W --fibers ${FIBERS:-131072} --variant $V --steps 2 --warmup 1 --no-cpu-baseline --no-solver"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pq1 -- python3 bench.py $A > /dev/null 2> $O/pq1.err; echo pq1 done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pq2 -- python3 bench.py $A > /dev/null 2> $O/pq2.err; echo pq2 done
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_MFMA --output-format csv -d $O/pq3 -- python3 bench.py $A > /dev/null 2> $O/pq3.err; echo pq3 done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("$O/pq1", "$O/pq2", "$O/pq3"):
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            m = re.search(r"(k_fiber_\w+)<", r["Kernel_Name"])
            if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/${TAG}_${W}_sq.txt", "w") as f:
    for k in acc:
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        print(k, "mean per dispatch over", len(next(iter(acc[k].values()))), "dispatches", file=f)
        for n in sorted(c): print(f"  {n:32s} {c[n]:.4e}", file=f)
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
                if n in c: print(f"  share of wave cycles {n:24s} {c[n] / wc:.3f}", file=f)
print(open("$O/${TAG}_${W}_sq.txt").read())
PY
find $O/pq1 $O/pq2 $O/pq3 -name "*.csv" -size +2M -delete
