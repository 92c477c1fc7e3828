"""Oracle-fed replay of the reference's Test_bellman_pi3d (test/transition_prob/tprob_test.c:2448-2540): 3 states, 3 controls,
drift f3, diffusion I, stagecost3d, boundcost 100, box [-1,2] x [-2,3] x [-3,1], 25^3 nodes, every face absorbing, discount 0.1,
fixed rank 10 (ApproxArgs 1e-8 / 1e-7 / kick 10 / no adaptation / start 10 / max 10), start value |x|^2; control updates of
pi_solve(20 sweeps, 1e-3) + one vi_solve step until |V_vi - V_pi| < 1e-3 (at most 400).  The reference minimises over
u in [-5,5]^3 with C3's BFGS; the oracle path scans a 5 x 5 x 5 candidate list over the same box.  Writes history and the final
value function (tests/golden/closed_loop_pi3d_oracle.npz).

    python tools/run_reference_pi3d.py oracle|gpu out.npz [max_updates]"""
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
    max_updates = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    t0 = time.time()
    loop = CL.pi3d_loop(path)

    def progress(ii, cost):
        if ii % 10 == 0:
            _, diff, norm, rank = loop.history[-1]
            print(f"update {ii:4d}  |V_vi-V_pi| {diff:.6e}  |V| {norm:.9f}  rank {rank}  sweeps {loop.sweeps}  {time.time() - t0:7.1f} s", flush=True)

    cost = loop.run(max_updates=max_updates, on_update=progress)
    ranks, cores = loop.cores_of(cost)
    norm = loop.norm(cost)
    print(f"Test_bellman_pi3d via {path}: {len(loop.history)} control updates, {loop.sweeps} sweeps in {time.time() - t0:.1f} s, |V| = {norm:.9f}, "
          f"last |V_vi-V_pi| = {loop.history[-1][1]:.3e}", flush=True)
    np.savez_compressed(out, path=path, history=np.array(loop.history), ranks=np.array(ranks), core0=cores[0], core1=cores[1], core2=cores[2],
                        norm=norm, cands=loop.w.cands, sweeps=loop.sweeps, seconds=time.time() - t0)


if __name__ == "__main__":
    main()
