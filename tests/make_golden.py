"""Generate tests/golden/car7d_small.npz from the pinned CPU oracle (run once; output committed).
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
