"""Oracle-fed replay of the reference's closed-loop test Test_bellman_vi (test/transition_prob/tprob_test.c:1817-1897):
value iteration ONLY (c3control_vi_solve, 10 000 sweeps or until the step falls below 1e-5) on the 2-D problem of the
regression (drift (x1, u), diffusion I, stage x0^2 + x1^2 + u^2, boundcost 100, [-2,2]^2 reflecting, discount 0.1, 100 x 100
nodes, start value 0.2; ApproxArgs 1e-8 / 1e-8 / kick 5 / adapt / start rank 2 / maxrank 30 -- clamped to 20, the largest rank
the device kernels serve, on both paths) with u in [-3, 3].  The reference minimises with C3's BFGS; the oracle path scans a
49-point candidate list over the same box.  Writes the final value function and the history to an .npz
(tests/golden/closed_loop_vi_oracle.npz was made with: python tools/run_reference_closed_loop.py oracle <out>; about 3 minutes
of one core).

    python tools/run_reference_closed_loop.py oracle|gpu out.npz [max_sweeps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import closed_loop_lib as CL  # noqa: E402


def main():
    path, out = sys.argv[1], sys.argv[2]
    max_sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    t0 = time.time()
    loop = CL.vi_loop(path)
    cost, hist = CL.vi_solve_logged(loop, max_sweeps, 1e-5, every=max(1, max_sweeps // 40))
    ranks, cores = loop.cores_of(cost)
    norm = loop.norm(cost)
    print(f"Test_bellman_vi via {path}: {len(hist)} sweeps in {time.time() - t0:.1f} s, |V| = {norm:.9f}, last step {hist[-1][1]:.3e}, rank {ranks[1]}", flush=True)
    np.savez_compressed(out, path=path, history=np.array(hist), ranks=np.array(ranks), core0=cores[0], core1=cores[1], nodal=loop.nodal(cost),
                        norm=norm, cands=loop.w.cands, seconds=time.time() - t0)


if __name__ == "__main__":
    main()
