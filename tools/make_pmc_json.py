"""Aggregate rocprofv3 --pmc CSVs (separate FETCH_SIZE / WRITE_SIZE / SQ passes) into profiles/<name>_pmc.json.
usage: python tools/make_pmc_json.py out.json dir1 dir2 ...   (per-dispatch averages per kernel)"""
import collections, csv, glob, json, re, sys

import os

NODES = int(os.environ.get("C3SC_PMC_NODES", 5373952))  # node backups per launch of the profiled run (F x N)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            m = re.search(r"(k_fiber_\w+<[^>]*>)", r["Kernel_Name"])
            if m:
                acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_what": "rocprofv3 --pmc passes (separate runs for FETCH_SIZE, WRITE_SIZE and the SQ set) of `python3 bench.py --steps 2 "
                "--warmup 1 --no-cpu-baseline`, per-dispatch averages per kernel; FETCH/WRITE_SIZE are KiB counters, FETCH is "
                "reported raw and with the x2 gfx950 wide-read correction (MI355X_MICROARCH.md HBM section; these reads are not a "
                "wide coalesced stream, so the correction is an upper bound)",
       "nodes_per_launch": NODES, "fibers_per_dim": int(os.environ.get("C3SC_PMC_FIBERS", 0)), "kernels": {}}
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    if "FETCH_SIZE" in c:
        c["fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024
        c["fetch_bytes_x2_gfx950_correction"] = 2 * c["fetch_bytes_per_launch"]
    if "WRITE_SIZE" in c:
        c["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_node_uncorrected"] = (c["fetch_bytes_per_launch"] + c["write_bytes_per_launch"]) / NODES
    if "SQ_INSTS_VALU_FMA_F64" in c:
        fl = 64 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0))
        mf = 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)  # one v_mfma_f64_16x16x4_f64 = 4 MOPS = 2048 flop
        c["executed_valu_f64_flop_per_launch"] = fl
        c["executed_mfma_f64_flop_per_launch"] = mf
        c["executed_flop_per_node"] = (fl + mf) / NODES
        c["executed_flops_per_node"] = (fl + mf) / NODES
        c["executed_mfma_share"] = mf / (fl + mf) if fl + mf > 0 else 0.0
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
        c["valu_active_share_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in c:
                c[n.lower() + "_share_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
        c["lds_bank_conflict_share_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][k] = c
json.dump(out, open(sys.argv[1], "w"), indent=1)
ks = out["kernels"].values()
print("mean fetch MB", sum(c.get("fetch_bytes_per_launch", 0) for c in ks) / len(ks) / 1e6,
      "write MB", sum(c.get("write_bytes_per_launch", 0) for c in ks) / len(ks) / 1e6,
      "flop/node", sum(c.get("executed_flop_per_node", 0) for c in ks) / len(ks),
      "valu share", sum(c.get("valu_active_share_of_wave_cycles", 0) for c in ks) / len(ks))
