"""Generate tests/golden/car7d_small.npz and tests/golden/pi_dubins_small.npz from the pinned CPU oracle (run once;
output committed).
The fixture is data only: seeded inputs are regenerated from c3sc_amd.workloads, the file holds the
fiber indices and the oracle's outputs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib  # noqa: E402
from c3sc_amd import workloads as wl  # noqa: E402

oracle_lib.build()
ngrid, rank = (9, 8, 10, 7, 6, 5, 11), 4
w = wl.c4_car7d().scaled(ngrid=ngrid, rank=rank)
P = oracle_lib.Problem(w, wl.synth_cores(w))
data = {"ngrid": np.array(ngrid), "rank": np.array(rank)}
for k in range(w.dx):
    idx = wl.synth_fibers(w, k, 16, seed=0x601D)
    idx[0, :] = 0
    idx[1, :] = np.array(ngrid) - 1
    idx[:, k] = 0
    out, ui, ab = P.bellman_fibers(k, idx)
    data[f"idx{k}"], data[f"out{k}"], data[f"ui{k}"], data[f"ab{k}"] = idx, out, ui, ab
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "car7d_small.npz"), **data)
print("wrote car7d_small.npz")


# policy evaluation (bellman_pi, bellman.c:1702-1886): policy greedy for one value function, right-hand side on another
ngrid2, rank2 = (21, 17, 16), 4
w2 = wl.c2_dubins().scaled(ngrid=ngrid2, rank=rank2)
cores_pol = wl.synth_cores(w2)
cores_it = wl.smooth_cores(w2)
P2 = oracle_lib.Problem(w2, cores_it)
pol_vf = oracle_lib.ValueF(w2.ngrid, w2.ranks, cores_pol)
Ppol = oracle_lib.Problem(w2, cores_pol)
data2 = {"ngrid": np.array(ngrid2), "rank": np.array(rank2)}
for k in range(w2.dx):
    idx = wl.synth_fibers(w2, k, 12, seed=0x601E)
    idx[0, :] = 0
    idx[1, :] = np.array(ngrid2) - 1
    idx[:, k] = 0
    _, ui, ab = Ppol.bellman_fibers(k, idx)          # the greedy policy (candidate index per node)
    P2.pi_begin()                                     # fresh tables: nothing cached, every policy is computed here
    P2.pi_step_begin()
    out, ui2 = P2.policy_fibers(pol_vf, k, idx)
    assert ((ui2 == ui) | (ui2 < 0)).all()
    data2[f"idx{k}"], data2[f"policy{k}"], data2[f"out{k}"], data2[f"ab{k}"] = idx, ui, out, ab
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pi_dubins_small.npz"), **data2)
print("wrote pi_dubins_small.npz")
