"""Runs one case of the reference's end-to-end regression (tests/regression_lib.py: tprob_test.c:1996-2357) and writes
the history and the final value function to an .npz:

    python tools/run_reference_regression.py pi_100 oracle out.npz [max_updates]     # CPU oracle fibers
    python tools/run_reference_regression.py pi_100 gpu out.npz [max_updates] [bfgs] # libc3sc.so, device fibers

The oracle run of pi_50 / pi_100 is what tests/golden/regression_*.npz were made with (hours of one CPU core for pi_100:
the reference's string memo and its FT evaluation before every memo lookup are kept)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import regression_lib as R  # noqa: E402


def main():
    case, path, out = sys.argv[1], sys.argv[2], sys.argv[3]
    max_updates = int(sys.argv[4]) if len(sys.argv) > 4 else None
    minimiser = sys.argv[5] if len(sys.argv) > 5 else "bruteforce"
    loop = R.OracleLoop(case) if path == "oracle" else R.GpuLoop(case, minimiser)
    t0 = time.time()
    every = max(1, (max_updates or loop.max_updates) // 50)

    def progress(ii, cost):
        if ii % every == 0:
            _, diff, norm, rank = loop.history[-1]
            print(f"update {ii:6d}  |V_vi-V_pi| {diff:.6e}  |V| {norm:.9f}  rank {rank:2d}  sweeps {loop.sweeps}  {time.time() - t0:8.1f} s", flush=True)

    cost = loop.run(max_updates=max_updates, on_update=progress)
    norm = loop.norm(cost)
    ranks, cores = loop.cores_of(cost)
    hist = np.array(loop.history, dtype=np.float64)
    print(f"{case} via {path}/{minimiser}: {len(loop.history)} updates, {loop.sweeps} sweeps, {time.time() - t0:.1f} s; |V| = {norm:.12f}, "
          f"anchor |100-|V||/100 = {R.anchor(norm):.4f}, last |V_vi-V_pi| = {loop.history[-1][1]:.3e}, rank {ranks[1]}", flush=True)
    np.savez_compressed(out, case=case, path=path, minimiser=minimiser, history=hist, ranks=np.array(ranks), core0=cores[0], core1=cores[1],
                        nodal=loop.nodal(cost), norm=norm, sweeps=loop.sweeps, seconds=time.time() - t0)


if __name__ == "__main__":
    main()
