"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_summary.py <dir> [name-filter]"""
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
