"""Do the rocprofv3 kernel stats reproduce the bench line?   python tools/check_kernel_stats.py <kernel_stats.csv> <bench.json>

Prints, for every k_fiber_* row of the CSV, its AverageNs next to roofline.avg_launch_ms of the JSON line the same (profiled)
run printed, the mean over the rows, and the roofline fraction that mean gives; exits non-zero if the mean is off by more than 5 %."""
import csv
import json
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fiber_" in r["Name"]]
line = [l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1]
rf = json.loads(line)["roofline"]
ref_all = rf["avg_launch_ms"]
by_dim = rf.get("launch_ms_by_dim")  # HIP events per varying dimension = per kernel instantiation <Model, RP, K, ...>
tot_ns, calls, worst = 0.0, 0, 0.0
for r in rows:
    name = r["Name"].replace("c3sc::", "").replace("void ", "")
    name = name[: name.index("(")] if "(" in name else name
    avg = float(r["AverageNs"]) * 1e-6
    ref_ms = ref_all
    m = re.search(r"<.*?>?, (\d+), (\d+)[,>]", name)
    if by_dim and m and int(m.group(2)) < len(by_dim):
        ref_ms = by_dim[int(m.group(2))]
    worst = max(worst, abs(avg / ref_ms - 1.0))
    tot_ns += float(r["TotalDurationNs"])
    calls += int(r["Calls"])
    print(f"{name:60s} calls {int(r['Calls']):3d}  avg {avg:8.4f} ms  min {float(r['MinNs']) * 1e-6:8.4f}  max {float(r['MaxNs']) * 1e-6:8.4f}  "
          f"vs HIP events {ref_ms:8.4f} ms ({100.0 * (avg / ref_ms - 1.0):+5.1f} %)")
mean = tot_ns / max(calls, 1) * 1e-6
frac = rf["algorithmic_flops_per_node"] * rf["nodes_per_launch"] / (mean * 1e-3) / 1e12 / rf["peak"]
print(f"mean over {calls} dispatches: {mean:.4f} ms per launch (HIP events of the same run: {ref_all:.4f} ms, {100.0 * (mean / ref_all - 1.0):+.1f} %; worst row "
      f"{100.0 * worst:.1f} % off its own dimension's events); "
      f"roofline fraction from the CSV {frac:.3f}, from the JSON line {rf['frac']:.3f}")
sys.exit(0 if abs(mean / ref_all - 1.0) <= 0.05 and (by_dim is None or worst <= 0.05) else 1)
