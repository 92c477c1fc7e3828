"""tools/dense_truth.py restates the Bellman backup of the 7-D car densely in torch (the ground truth the TT solver's error is
measured against).  Here: that restatement against the C oracle on a small grid -- every node, flags included."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from c3sc_amd import workloads as wl  # noqa: E402


def test_dense_backup_matches_the_oracle_on_every_node(oracle):
    import torch

    import dense_truth as DT

    w = wl.c4_car7d().scaled(ngrid=(7, 6, 8, 5, 5, 4, 6), rank=3)
    cores = wl.smooth_cores(w, coef=[0.3, 0.5, 0.2, 0.1, 0.15, 0.7, 0.25])
    P = oracle.Problem(w, cores, consistent_ends=True)
    k = 2
    dims = [range(n) if m != k else [0] for m, n in enumerate(w.ngrid)]
    idx = np.array(list(itertools.product(*dims)), dtype=np.int32)
    out, _, ab = P.bellman_fibers(k, idx)
    shp = [n for m, n in enumerate(w.ngrid) if m != k] + [w.ngrid[k]]
    ref = np.moveaxis(out.reshape(shp), -1, k)
    dev = torch.device("cpu")
    V0 = DT.tt_to_dense(w.ngrid, w.ranks, [c.reshape(-1) for c in cores], torch, dev)
    got = DT.DenseCar7D(w, dev).apply(V0).numpy()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    flags = np.moveaxis(ab.reshape(shp), -1, k)
    op = DT.DenseCar7D(w, dev)
    np.testing.assert_array_equal(op.absorbed.numpy(), flags == 1)
    np.testing.assert_array_equal(op.inobs.numpy(), flags == -1)
