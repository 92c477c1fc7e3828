"""Own TT-cross driver behind valuef_interp (c3sc_amd/host/c3sc_cross.c).  The reference delegates
this to C3 and pins nothing numerically, so the tests are properties: nodal accuracy on functions of known TT
rank, rank adaptation + rounding, warm start, and the continuous L2 norms against closed forms."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FIBER_FN = C.CFUNCTYPE(C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def _lib():
    import facade_lib

    L = facade_lib.lib()
    L.valuef_interp.restype = C.c_void_p
    L.valuef_interp_batch.restype = C.c_void_p
    L.valuef_norm.restype = C.c_double
    L.valuef_norm2diff.restype = C.c_double
    L.valuef_eval.restype = C.c_double
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_eval_ind.restype = C.c_double
    return L, facade_lib


def _interp(L, fl, func, grids, vref=None, batch=False, **kw):
    d = len(grids)
    N = np.array([len(g) for g in grids], dtype=np.uintp)
    gs = [fl.f64(g) for g in grids]
    gp = fl.ptrs(gs)
    calls = {"fibers": 0, "calls": 0}

    def one(n, x, out, args):
        X = np.ctypeslib.as_array(x, shape=(n, d))
        np.ctypeslib.as_array(out, shape=(n,))[:] = func(X)
        calls["fibers"] += 1
        calls["calls"] += 1
        return 0

    def many(F, n, x, out, args):
        X = np.ctypeslib.as_array(x, shape=(F * n, d))
        np.ctypeslib.as_array(out, shape=(F * n,))[:] = func(X)
        calls["fibers"] += F
        calls["calls"] += 1
        return 0

    aa = C.c_void_p(L.approx_args_init())
    L.approx_args_set_cross_tol(aa, C.c_double(kw.get("cross_tol", 1e-10)))
    L.approx_args_set_round_tol(aa, C.c_double(kw.get("round_tol", 1e-10)))
    L.approx_args_set_kickrank(aa, C.c_size_t(kw.get("kickrank", 2)))
    L.approx_args_set_startrank(aa, C.c_size_t(kw.get("startrank", 2)))
    L.approx_args_set_maxrank(aa, C.c_size_t(kw.get("maxrank", 12)))
    L.approx_args_set_adapt(aa, C.c_int(kw.get("adapt", 1)))
    if "crossrank" in kw:
        L.approx_args_set_crossrank(aa, C.c_size_t(kw["crossrank"]))
    if "cross_maxiter" in kw:
        L.approx_args_set_cross_maxiter(aa, C.c_size_t(kw["cross_maxiter"]))
    if batch:
        cb = BATCH_FN(many)
        vf = C.c_void_p(L.valuef_interp_batch(C.c_size_t(d), cb, None, fl.sp(N), gp, vref, aa, 0))
    else:
        cb = FIBER_FN(one)
        vf = C.c_void_p(L.valuef_interp(C.c_size_t(d), cb, None, fl.sp(N), gp, vref, aa, 0))
    L.approx_args_free(aa)
    ranks = [L.valuef_get_ranks(vf)[i] for i in range(d + 1)]
    return vf, ranks, calls


def _nodal_error(L, fl, vf, func, grids, nsamp=400, seed=1):
    rng = np.random.default_rng(seed)
    d = len(grids)
    worst = 0.0
    for _ in range(nsamp):
        ind = np.array([rng.integers(0, len(g)) for g in grids], dtype=np.uintp)
        x = np.array([[grids[m][ind[m]] for m in range(d)]])
        worst = max(worst, abs(L.valuef_eval_ind(vf, fl.sp(ind)) - func(x)[0]))
    return worst


def test_cross_exact_low_rank_and_rounding():
    L, fl = _lib()
    grids = [np.linspace(-1, 2, 17), np.linspace(-2, 3, 21), np.linspace(-3, 1, 15), np.linspace(0, 1, 19)]
    quad = lambda X: (X ** 2).sum(axis=1) + 0.5  # TT rank 2
    vf, ranks, calls = _interp(L, fl, quad, grids, startrank=4, maxrank=10)
    assert ranks[0] == ranks[-1] == 1 and max(ranks) == 2, ranks  # rounding trims the start rank 4 to the true rank
    assert _nodal_error(L, fl, vf, quad, grids) < 1e-9
    # the batched callback sees all fibers of a core step in one call
    vfb, ranksb, callsb = _interp(L, fl, quad, grids, startrank=4, maxrank=10, batch=True)
    assert ranksb == ranks and callsb["fibers"] == calls["fibers"] and callsb["calls"] < calls["calls"] / 4
    L.valuef_destroy(vf)
    L.valuef_destroy(vfb)


def test_cross_rank_adaptation_and_warm_start():
    L, fl = _lib()
    grids = [np.linspace(-1, 1, 25), np.linspace(-1, 1, 23), np.linspace(-1, 1, 21)]
    f = lambda X: 1.0 / (1.0 + (X ** 2).sum(axis=1))
    vf, ranks, calls = _interp(L, fl, f, grids, startrank=2, kickrank=2, maxrank=14, cross_tol=1e-8, round_tol=1e-7)
    assert max(ranks) > 2  # kicked above the start rank
    assert _nodal_error(L, fl, vf, f, grids) < 5e-6
    # warm start from the previous value function (ranks + 1, index sets copied): a slightly different function
    g = lambda X: 1.0 / (1.05 + (X ** 2).sum(axis=1))
    vf2, ranks2, calls2 = _interp(L, fl, g, grids, vref=vf, startrank=2, kickrank=2, maxrank=14, cross_tol=1e-8, round_tol=1e-7)
    assert _nodal_error(L, fl, vf2, g, grids) < 5e-6
    assert calls2["fibers"] <= calls["fibers"]  # no rank search from scratch
    d = L.valuef_norm2diff(vf, vf2)
    assert 0 < d < 0.2
    L.valuef_destroy(vf)
    L.valuef_destroy(vf2)


def test_elevated_cross_rank_rounded_to_the_cap_is_near_the_best_train():
    """approx_args_set_crossrank: a function whose TT ranks exceed the cap (a kinked, non-separable function -- the value functions
    of exit-time problems look like this).  Interpolation through maxrank fibers per core is a multiple of the best rank-capped
    train's error away; the cross approximation at 2x / 3x the cap, cut back to the cap by the TT-SVD, must be within 1.5x of the
    TT-SVD truncation of the full tensor and better than plain interpolation at the cap."""
    L, fl = _lib()
    L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
    n, d, cap = 13, 4, 5
    grids = [np.linspace(-1.0, 1.0, n) + 0.01 * m for m in range(d)]
    f = lambda X: np.abs(X[:, 0] + 0.7 * X[:, 1] - 0.5 * X[:, 2] + 0.3 * X[:, 3]) + 0.2 * np.sqrt(0.1 + (X ** 2).sum(axis=1))
    mesh = np.stack(np.meshgrid(*grids, indexing="ij"), axis=-1).reshape(-1, d)
    full = f(mesh).reshape((n,) * d)

    def ttsvd_err(T, r):
        A, r0, cores = T.copy(), 1, []
        for m in range(d - 1):
            A = A.reshape(r0 * n, -1)
            U, S, Vt = np.linalg.svd(A, full_matrices=False)
            rr = min(r, len(S))
            cores.append(U[:, :rr].reshape(r0, n, rr))
            A, r0 = S[:rr, None] * Vt[:rr], rr
        cores.append(A.reshape(r0, n, 1))
        acc = cores[0]
        for G in cores[1:]:
            acc = np.tensordot(acc, G, axes=([acc.ndim - 1], [0]))
        return np.linalg.norm(acc.reshape(T.shape) - T) / np.linalg.norm(T)

    def dense(vf, ranks):
        pp = L.valuef_get_cores(vf)
        acc = np.ones((1, 1))
        for m in range(d):
            G = np.ctypeslib.as_array(pp[m], shape=(n * ranks[m] * ranks[m + 1],)).reshape(n, ranks[m + 1], ranks[m]).transpose(2, 0, 1)
            acc = np.tensordot(acc, G, axes=([acc.ndim - 1], [0]))
        return acc.reshape((n,) * d)

    best = ttsvd_err(full, cap)
    errs = {}
    for cr in (0, 2 * cap, 3 * cap):
        vf, ranks, calls = _interp(L, fl, f, grids, startrank=cap, kickrank=cap, maxrank=cap, crossrank=cr, cross_tol=1e-8, round_tol=1e-8, batch=True)
        assert max(ranks) <= cap, ranks  # the result never exceeds maxrank, whatever the cross ran at
        errs[cr] = np.linalg.norm(dense(vf, ranks) - full) / np.linalg.norm(full)
        L.valuef_destroy(vf)
    print(f"best rank-{cap} train {best:.3e}; interpolation at the cap {errs[0]:.3e}; cross at {2 * cap} / {3 * cap} rounded to {cap}: "
          f"{errs[2 * cap]:.3e} / {errs[3 * cap]:.3e}")
    assert errs[2 * cap] < errs[0] and errs[3 * cap] < errs[0]
    assert errs[3 * cap] <= 1.5 * best


def test_elevated_cross_rank_edge_cases():
    """crossrank is clamped per bond to the smaller side of the unfolding (a 2-D function on 9 x 30 nodes cannot have a bond above 9),
    is ignored without rank adaptation (adapt = 0: no kicks, the start rank stays), is carried over by a warm start (the second
    interpolation starts from the first one's CROSS ranks and index sets, not from the rounded ranks + 1) and never lets the result
    exceed maxrank; an exactly low-rank function is still recovered to rounding accuracy."""
    L, fl = _lib()
    g2 = [np.linspace(-1, 1, 9), np.linspace(0, 2, 30)]
    f2 = lambda X: np.exp(-((X[:, 0] - 0.3 * X[:, 1]) ** 2)) + 0.1 * np.abs(X[:, 0] + X[:, 1] - 1.0)
    vf, ranks, calls = _interp(L, fl, f2, g2, startrank=2, kickrank=3, maxrank=4, crossrank=64, cross_tol=1e-10, round_tol=1e-10, batch=True)
    assert ranks == [1, 4, 1], ranks
    assert calls["fibers"] <= 2 * 5 * 3 * (9 + 30) * 2  # cross ranks stopped at 9, the smaller side
    vf2, ranks2, calls2 = _interp(L, fl, lambda X: f2(X) * 1.01, g2, vref=vf, startrank=2, kickrank=3, maxrank=4, crossrank=64, cross_tol=1e-10,
                                  round_tol=1e-10, batch=True)
    assert ranks2 == [1, 4, 1] and calls2["fibers"] < calls["fibers"]  # no rank search from the start rank again
    L.valuef_destroy(vf)
    L.valuef_destroy(vf2)
    grids = [np.linspace(-1, 2, 11), np.linspace(-2, 3, 12), np.linspace(0, 1, 10)]
    quad = lambda X: (X ** 2).sum(axis=1) + 0.5  # TT rank 2
    vf, ranks, _ = _interp(L, fl, quad, grids, startrank=3, maxrank=3, crossrank=9, adapt=0)
    assert max(ranks) <= 3 and _nodal_error(L, fl, vf, quad, grids) < 1e-9
    L.valuef_destroy(vf)
    vf, ranks, _ = _interp(L, fl, quad, grids, startrank=2, kickrank=2, maxrank=3, crossrank=9)
    assert max(ranks) == 2 and _nodal_error(L, fl, vf, quad, grids) < 1e-9  # rounding still trims to the true rank
    L.valuef_destroy(vf)


def test_cross_iteration_cap_and_warm_started_sequence():
    """approx_args_set_cross_maxiter: the reference caps the cross iterations of an interpolation at 5 (valuefunc.c:632).  With the cap
    at 1 a single interpolation from generic index sets is rough, but a SEQUENCE of warm-started interpolations of a slowly changing
    function -- what a value iteration is -- reaches the accuracy of the 5-iteration driver with a fraction of the fibers: the sweeps
    play the role of the cross iterations."""
    L, fl = _lib()
    grids = [np.linspace(-1, 1, 17), np.linspace(-1, 1, 15), np.linspace(-1, 1, 16), np.linspace(-1, 1, 14)]
    fam = lambda s: (lambda X: 1.0 / (1.0 + s + (X ** 2).sum(axis=1)) + 0.05 * np.abs(X[:, 0] - X[:, 3] + 0.1 * s))
    res = {}
    for cap in (5, 1):
        vf, fibers, err = None, 0, None
        for step in range(8):
            f = fam(0.02 * step)
            nxt, ranks, calls = _interp(L, fl, f, grids, vref=vf, startrank=6, kickrank=2, maxrank=6, cross_tol=1e-12, round_tol=1e-12,
                                        cross_maxiter=cap, batch=True)
            fibers += calls["fibers"]
            if vf is not None:
                L.valuef_destroy(vf)
            vf = nxt
            err = _nodal_error(L, fl, vf, f, grids, nsamp=600)
        res[cap] = (err, fibers)
        L.valuef_destroy(vf)
    print(f"8 warm-started interpolations of a drifting function: cap 5 -> error {res[5][0]:.2e} with {res[5][1]} fibers; cap 1 -> {res[1][0]:.2e} with {res[1][1]}")
    assert res[1][1] <= 0.8 * res[5][1]
    assert res[1][0] <= 2.0 * res[5][0] + 1e-12


def test_continuous_norms_and_offgrid_eval():
    L, fl = _lib()
    # constant 0.2 on [-2,2]^2 (tprob_test.c quad2d): ||V||_L2 = 0.2 * 4
    g2 = [np.linspace(-2, 2, 30), np.linspace(-2, 2, 26)]
    vf, ranks, _ = _interp(L, fl, lambda X: np.full(len(X), 0.2), g2)
    assert max(ranks) == 1
    assert L.valuef_norm(vf) == pytest.approx(0.8, rel=1e-12)
    # bilinear function: the multilinear interpolant is exact -> closed-form norm and off-grid values
    g3 = [np.linspace(0, 1, 9), np.linspace(0, 2, 12), np.linspace(-1, 1, 7)]
    bil = lambda X: (1 + X[:, 0]) * (2 - X[:, 1]) * (0.5 + X[:, 2])
    vb, _, _ = _interp(L, fl, bil, g3)
    exact = np.sqrt((7.0 / 3.0) * (8.0 / 3.0) * (2 * 0.25 + 2.0 / 3.0))
    assert L.valuef_norm(vb) == pytest.approx(exact, rel=1e-11)
    rng = np.random.default_rng(3)
    for _ in range(50):
        x = np.array([rng.uniform(0, 1), rng.uniform(0, 2), rng.uniform(-1, 1)])
        assert L.valuef_eval(vb, fl.dp(x)) == pytest.approx(bil(x[None, :])[0], rel=1e-12)
    # norm2diff resolves differences far below sqrt(eps)*norm (no Gram-matrix cancellation)
    vb2, _, _ = _interp(L, fl, lambda X: bil(X) * (1 + 1e-9), g3)
    assert L.valuef_norm2diff(vb, vb2) == pytest.approx(1e-9 * exact, rel=1e-4)
    for v in (vf, vb, vb2):
        L.valuef_destroy(v)


def test_rank_adaptation_survives_degenerate_index_sets(oracle):
    """T(V0) of the symmetric 2-D LQG problem has mirror-image rows and columns, so a cross step at a rank above the
    number of distinct ones factors an exactly rank-deficient matrix and picks pivots in null directions; rounding then
    drops a rank although the function needs more, and which way it goes depends on rounding noise in the fiber values.
    The driver must not stop there: with 1e-16 relative noise on the fiber values every run has to reach the rounding
    accuracy at all nodes (CPU only: fibers from the oracle)."""
    sys.path.insert(0, ROOT)
    from c3sc_amd import workloads as wl

    w = wl.c1_lqg2d().scaled(ngrid=(19, 17))
    import facade_lib as fl

    L = fl.lib()
    for n in ("valuef_interp", "valuef_create_nodal"):
        getattr(L, n).restype = C.c_void_p
    L.valuef_eval_ind.restype = C.c_double
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    cores = [np.full((w.ngrid[0], 1), 0.2), np.ones((w.ngrid[1], 1))]
    wr = wl.Workload(w.name, w.model, w.params, w.dx, w.du, w.lb, w.ub, w.ngrid, (1, 1, 1), w.discount, w.bc, list(w.obstacles), w.cands)
    P = oracle.Problem(wr, cores)
    idx = np.zeros((w.ngrid[1], 2), dtype=np.int32)
    idx[:, 1] = np.arange(w.ngrid[1])
    want = P.bellman_fibers(0, idx)[0].T  # T(V0) at every node
    xg = [np.linspace(w.lb[m], w.ub[m], w.ngrid[m]) for m in range(2)]
    gs = [fl.f64(g) for g in xg]
    gp = fl.ptrs(gs)
    Ng = np.array(w.ngrid, dtype=np.uintp)
    v0 = C.c_void_p(L.valuef_create_nodal(C.c_size_t(2), fl.sp(Ng), fl.sp(np.array([1, 1, 1], dtype=np.uintp)), fl.ptrs([fl.f64(c) for c in cores])))
    worst = 0.0
    for seed in range(8):
        rng = np.random.default_rng(seed)

        def fiber(n, x, out, a):
            X = np.ctypeslib.as_array(x, shape=(n, 2)).copy()
            v = P.bellman_vi(X, use_memo=False)[0]
            np.ctypeslib.as_array(out, shape=(n,))[:] = v * (1.0 + 1e-16 * rng.standard_normal(n))
            return 0

        cb = FIBER_FN(fiber)
        aa = C.c_void_p(L.approx_args_init())
        L.approx_args_set_cross_tol(aa, C.c_double(1e-10))
        L.approx_args_set_round_tol(aa, C.c_double(1e-9))
        L.approx_args_set_kickrank(aa, C.c_size_t(3))
        L.approx_args_set_startrank(aa, C.c_size_t(3))
        L.approx_args_set_maxrank(aa, C.c_size_t(17))
        vf = C.c_void_p(L.valuef_interp(C.c_size_t(2), cb, None, fl.sp(Ng), gp, v0, aa, 0))  # warm start from rank 1, as step_vi does
        got = np.array([[L.valuef_eval_ind(vf, fl.sp(np.array([i, j], dtype=np.uintp))) for j in range(w.ngrid[1])] for i in range(w.ngrid[0])])
        worst = max(worst, np.abs(got - want).max())
        L.valuef_destroy(vf)
        L.approx_args_free(aa)
    L.valuef_destroy(v0)
    print("worst nodal error over noise seeds:", worst)
    assert worst <= 2e-8 * np.abs(want).max()


_THREADS_SNIPPET = r"""
import ctypes as C, hashlib, sys
import numpy as np
sys.path.insert(0, {tests!r})
import test_cross_driver as t
L, fl = t._lib()
L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
n, d = 48, 4
grids = [np.linspace(-1.0, 1.0, n) + 0.01 * m for m in range(d)]
f = lambda X: np.abs(X[:, 0] + 0.7 * X[:, 1] - 0.5 * X[:, 2] + 0.3 * X[:, 3]) + 0.2 * np.sqrt(0.1 + (X ** 2).sum(axis=1))
vf, ranks, calls = t._interp(L, fl, f, grids, startrank=24, kickrank=8, maxrank=6, crossrank=24, cross_tol=1e-8, round_tol=1e-8, batch=True, cross_maxiter=2)
pp = L.valuef_get_cores(vf)
h = hashlib.sha256()
for m in range(d):
    h.update(np.ctypeslib.as_array(pp[m], shape=(n * ranks[m] * ranks[m + 1],)).tobytes())
print("RESULT", ranks, h.hexdigest())
rng = np.random.default_rng(5)
vals = []
for _ in range(40):
    ind = np.array([rng.integers(0, n) for _ in range(d)], dtype=np.uintp)
    vals.append(float(L.valuef_eval_ind(vf, fl.sp(ind))))
print("VALUES", " ".join(repr(v) for v in vals))
"""


def test_rounding_does_not_depend_on_the_number_of_host_threads():
    """The dense host loops of the rounding (Householder panels, columns of Q, small products: c3sc_cross.c) run over whole columns
    on a small thread pool when the matrices are large (an elevated cross rank: 1152 x 24 here).  Every column's arithmetic is the
    serial loop's, so the rounded train must be the same BITS with one thread and with four (C3SC_THREADS_KEEP: the pool stays on
    even where its start-up measurement would switch it off)."""
    import subprocess
    import sys
    outs = []
    for threads in ("1", "4"):
        env = dict(os.environ, C3SC_THREADS=threads, C3SC_THREADS_KEEP="1")
        r = subprocess.run([sys.executable, "-c", _THREADS_SNIPPET.format(tests=os.path.dirname(os.path.abspath(__file__)))],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1]
        outs.append(line)
    print(outs[0])
    assert outs[0] == outs[1], outs
    assert "[1, 6, 6, 6, 1]" in outs[0], outs[0]  # the cap binds: the cross ran at 24 and was cut back


def test_rounding_svd_bidiagonal_qr_against_jacobi():
    """The square factors of the rounding (24 x 24 here, 48 x 48 at car7d's cross rank) are decomposed by Householder
    bidiagonalisation + implicit-shift QR (svd_gkr, c3sc_cross.c); C3SC_JACOBI_SVD=1 keeps the one-sided Jacobi everywhere.  Both
    must round the same cross approximation to the same ranks and the same function (values at 40 nodes to 1e-10 of the largest)."""
    import subprocess
    import sys
    res = []
    for jac in (False, True):
        env = dict(os.environ, C3SC_THREADS="1")
        env.pop("C3SC_JACOBI_SVD", None)
        if jac:
            env["C3SC_JACOBI_SVD"] = "1"
        r = subprocess.run([sys.executable, "-c", _THREADS_SNIPPET.format(tests=os.path.dirname(os.path.abspath(__file__)))],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        ranks = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split("]")[0]
        vals = np.array([float(x) for x in [l for l in r.stdout.splitlines() if l.startswith("VALUES")][-1].split()[1:]])
        res.append((ranks, vals))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    diff = np.abs(res[0][1] - res[1][1]).max() / np.abs(res[1][1]).max()
    print(f"bidiagonal QR vs Jacobi in the rounding: {res[0][0]}], largest difference of 40 node values {diff:.2e} of the largest value")
    assert diff < 1e-10
