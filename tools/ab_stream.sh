cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export C3SC_CROSS_MAXITER=1 C3SC_CROSS_RANK_FACTOR=${XRF:-1}
for rep in 1 2 3; do for mr in ${MRS:-16 0}; do C3SC_STREAM_MIN_RANK=$mr timeout -k 10 100 python tools/vi_sweep_quick.py car7d 80 > gpurun_out/ab_stream_$mr.txt 2>&1; python - <<PY
import re,statistics
v=[float(re.search(r"sweep\s+\d+:\s+([0-9.]+) ms",l).group(1)) for l in open("gpurun_out/ab_stream_$mr.txt") if l.startswith("sweep")]
print("min rank $mr: median of sweeps 20-79 %.3f ms, mean %.3f"%(statistics.median(v[20:]),statistics.mean(v[20:])))
PY
done; done
