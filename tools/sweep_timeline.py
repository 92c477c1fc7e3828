"""Summarises a rocprofv3 kernel trace (rocpd sqlite database) of tools/vi_sweep_quick.py: per-kernel statistics and the
timeline of one steady-state cross iteration (start offsets and durations in microseconds).
    python tools/sweep_timeline.py <results.db> > profiles/rNN_vi_sweep_car7d_timeline.txt"""
import collections
import sqlite3
import sys



def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("c3sc::", "")
    if n.startswith("void "):
        n = n[5:]
    depth, out = 0, []
    for ch in n:  # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[:90]


db = sqlite3.connect(sys.argv[1])
rows = [(short(n), s, e) for n, s, e in db.execute("select name, start, end from kernels order by start")]
agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[n].append((e - s) / 1e3)
print(f"{len(rows)} kernel dispatches")
print(f"{'kernel':90s} {'calls':>6s} {'mean us':>9s} {'total ms':>9s}")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:90s} {len(v):6d} {sum(v) / len(v):9.2f} {sum(v) / 1e3:9.2f}")
idx = [i for i, (n, s, e) in enumerate(rows) if "k_cross_idx" in n]
if len(idx) > 8:
    st, en = idx[-6], idx[-5]
    t0 = rows[st][1]
    print("\none cross iteration of a steady-state sweep (start offset, duration, kernel):")
    for n, s, e in rows[st:en]:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  {n}")
    gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(st, min(en, st + 28))]
    print(f"\nidle time between consecutive kernels inside the iteration: median {sorted(gaps)[len(gaps) // 2]:.2f} us, max {max(gaps):.2f} us")
