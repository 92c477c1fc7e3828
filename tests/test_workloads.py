"""Workload generators are deterministic and the algorithmic-work figures match SURVEY.md 8d."""
import numpy as np
import pytest

from c3sc_amd import workloads as wl


def test_splitmix64_reference_values():
    # published splitmix64 test vector: seed 1234567 -> first outputs
    z = wl.splitmix64(1234567, 3)
    assert [int(v) for v in z] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_synth_inputs_deterministic():
    w = wl.c4_car7d(n=9, r=3)
    a, b = wl.synth_cores(w), wl.synth_cores(w)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
        assert x.min() >= 0.3 and x.max() < 0.4
    f = wl.synth_fibers(w, 3, 100)
    assert f.dtype == np.int32 and f.shape == (100, 7) and (f[:, 3] == 0).all()
    assert (f >= 0).all() and (f < 9).all()
    cb = wl.cross_batch_fibers(w, 3)
    assert cb.shape == (9, 7)


@pytest.mark.parametrize("ctor,W,tol", [(wl.c2_dubins, 352, 2), (wl.c3_lqg6d, 3701, 2), (wl.c4_car7d, 2592, 2),
                                         (wl.c5_quad10d, 5471, 2)])
def test_algorithmic_flops_match_survey(ctor, W, tol):
    assert abs(wl.algorithmic_flops_per_node(ctor()) - W) <= tol


def test_smooth_cores_are_exact():
    w = wl.c2_dubins(n=7, r=3)
    cores = wl.smooth_cores(w, coef=[1.0, 2.0, 3.0])
    xg = w.xgrid()
    # brute-force contraction at a few nodes
    for ind in [(0, 0, 0), (3, 5, 6), (6, 1, 2)]:
        v = np.ones(1)
        for m in range(3):
            G = cores[m][ind[m]].reshape(w.ranks[m + 1], w.ranks[m]).T
            v = v @ G
        want = sum(c * xg[m][ind[m]] ** 2 for m, c in enumerate([1.0, 2.0, 3.0]))
        assert abs(v[0] - want) < 1e-13


def test_bench_cpu_baseline_worker_runs_without_gpu():
    """bench.py's cpu_baseline leg: one worker process (oracle, no torch / GPU) returns its node count and time."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--cpu-worker", "car7d", "0.5", "3"], timeout=120)
    r = json.loads(out.decode().strip().splitlines()[-1])
    assert r["nodes"] > 0 and r["nodes"] % 41 == 0 and 0.5 <= r["seconds"] < 30 and r["chunks"] >= 1
