# host threads of the rounding at cross rank 48 (one box, alternating): median sweep time and the rounding's own breakdown
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export C3SC_CROSS_MAXITER=1 C3SC_CROSS_RANK_FACTOR=4.8 C3SC_PROFILE=1
for rep in 1 2; do for t in 1 4 8 16; do
  C3SC_THREADS=$t timeout -k 10 100 python tools/vi_sweep_quick.py car7d 120 > gpurun_out/ab_threads_$t.txt 2>&1
  python - <<PY
import re,statistics
v=[float(re.search(r"sweep\s+\d+:\s+([0-9.]+) ms",l).group(1)) for l in open("gpurun_out/ab_threads_$t.txt") if l.startswith("sweep")]
r=[l for l in open("gpurun_out/ab_threads_$t.txt") if "rounding profile" in l][-1].split("checks): ")[1].strip()
print("threads $t: median of sweeps 60-119 %.3f ms | %s"%(statistics.median(v[60:]),r))
PY
done; done
