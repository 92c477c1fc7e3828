"""The fiber-quad kernel (kernel_fiber_quad.hpp: 16 fibers per wavefront, rank quarters in lane groups, f64 MFMA for the
varying-core products, in-register transposing reductions) against the CPU oracle: every varying dimension, ragged tiles,
faces, periodic wrap, obstacles, policy evaluation; values to 1e-12 of the value scale, flags bit-exact."""
import numpy as np
import pytest

from c3sc_amd import workloads as wl

pytestmark = pytest.mark.gpu
REL_TOL = 1e-12
QUAD = 4


def _engine(w, cores):
    from c3sc_amd.engine import BellmanEngine

    eng = BellmanEngine(0)
    eng.set_variant(QUAD)  # before the value is uploaded: the padded rank follows the variant (multiples of 4)
    eng.configure(w, cores)
    return eng


CASES = [
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), 300),
    ("car7d", dict(ngrid=(11, 12, 9, 13, 10, 11, 12), rank=10), 300),
    ("car7d", dict(), 64),                                                     # C4 at full size: 41^7, rank 10 (padded 12)
    ("quad10d", dict(ngrid=(5, 6, 5, 4, 5, 6, 5, 4, 5, 6), rank=4), 100),
    ("quad10d", dict(ngrid=(7, 6, 5, 8, 7, 6, 5, 8, 7, 25), rank=15), 70),     # C5's rank (padded 16)
    ("quad10d", dict(), 48),                                                   # C5 at FULL size: 25^10, rank 15 -- the benched launch's LDS layout (158 of 160 KB)
    ("quad10d", dict(ngrid=(6, 25, 7, 5, 9, 6, 24, 5, 8, 7), rank=13), 150),   # ragged grid, three tiles of the duo kernel, last one partial
    ("scar4d", dict(ngrid=(12, 11, 10, 9), rank=8), 200),
    ("scar4d", dict(), 50),                                                    # 40^4, rank 20: two MFMA row blocks
    ("lqg6d", dict(ngrid=(7, 8, 9, 6, 5, 7), rank=8), 150),
    ("lqg6d", dict(), 40),                                                     # C3: 31^6, rank 8, discounted scan
]


@pytest.mark.parametrize("name,kw,nf", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_fiber_quad_vs_oracle(oracle, name, kw, nf):
    w = wl.WORKLOADS[name]().scaled(**kw) if kw else wl.WORKLOADS[name]()
    cores = wl.synth_cores(w)
    P = oracle.Problem(w, cores)
    eng = _engine(w, cores)
    worst = 0.0
    for k in range(w.dx):
        idx = wl.synth_fibers(w, k, nf)
        idx[0, :] = 0
        idx[1, :] = np.array(w.ngrid) - 1
        idx[2, :] = 1
        idx[:, k] = 0
        ref, ref_ui, ref_ab = P.bellman_fibers(k, idx)
        out, ui, ab = eng.bellman_fibers_host(k, idx)
        assert "fiber_quad" in eng.last_kernel(), eng.last_kernel()
        if name == "quad10d" and w.ranks[1] > 12:  # rank class 16 at d = 10: two wavefronts per 16 fibers
            assert "fiber_quad_duo" in eng.last_kernel(), eng.last_kernel()
        assert eng.status() == 0
        np.testing.assert_array_equal(ab, ref_ab)
        scale = np.abs(ref).max()
        err = np.abs(out - ref).max()
        assert err <= REL_TOL * scale, f"{w.name} k={k}: err {err:.3e} scale {scale:.3e}"
        bad = ui != ref_ui
        assert not bad.any() or np.abs(out - ref)[bad].max() <= REL_TOL * scale
        worst = max(worst, err / scale)
    print(f"{name} {w.ngrid} rank {w.ranks[1]}: {eng.last_kernel()} max rel err {worst:.2e}")


def test_fiber_quad_policy_evaluation_and_repeatability(oracle):
    w = wl.c4_car7d().scaled(ngrid=(11,) * 7, rank=10)
    cores_pol = wl.synth_cores(w)
    cores_it = [c * (1.0 + 0.05 * np.cos(np.arange(c.size)).reshape(c.shape)) for c in wl.smooth_cores(w)]
    P = oracle.Problem(w, cores_it)
    pol_vf = oracle.ValueF(w.ngrid, w.ranks, cores_pol)
    eng_pol, eng_it = _engine(w, cores_pol), _engine(w, cores_it)
    P.pi_begin()
    P.pi_step_begin()
    for k in (0, 3, 6):
        idx = wl.synth_fibers(w, k, 90)
        ref, ref_ui = P.policy_fibers(pol_vf, k, idx)
        _, ui, _ = eng_pol.bellman_fibers_host(k, idx)
        out, _ = eng_it.policy_fibers_host(k, idx, ui)
        out2, _ = eng_it.policy_fibers_host(k, idx, ui)
        assert np.array_equal(out, out2)  # fixed association in the lane reductions: bit-identical repeats
        same = (ref_ui < 0) | (ui == ref_ui)
        assert np.abs(out - ref)[same].max() <= REL_TOL * np.abs(ref).max()
        assert "fiber_quad" in eng_it.last_kernel()


def test_duo_kernel_declines_large_grids_and_the_quad_kernel_takes_over(oracle):
    """The duo kernel keeps two staging buffers: at rank class 16 they hold cores of N <= 25 nodes.  A launcher that does not fit
    declines without launching and the next instantiation of the same padded rank runs (here the one-buffer quad kernel)."""
    w = wl.WORKLOADS["quad10d"]().scaled(ngrid=(5, 6, 30, 4, 5, 6, 5, 4, 5, 7), rank=14)
    cores = wl.synth_cores(w)
    P = oracle.Problem(w, cores)
    eng = _engine(w, cores)
    for k in (0, 2, 9):
        idx = wl.synth_fibers(w, k, 40)
        idx[:, k] = 0
        ref, _, ref_ab = P.bellman_fibers(k, idx)
        out, _, ab = eng.bellman_fibers_host(k, idx)
        # with dimension 2 varying its own core is not staged: everything else fits the duo kernel
        assert ("k_fiber_quad_duo<" if k == 2 else "k_fiber_quad<") in eng.last_kernel(), eng.last_kernel()
        np.testing.assert_array_equal(ab, ref_ab)
        assert np.abs(out - ref).max() <= REL_TOL * np.abs(ref).max()
