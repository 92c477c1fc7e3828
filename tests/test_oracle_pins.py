"""Pin the CPU oracle (oracle/c3sc_oracle.c) BEFORE it is trusted as the checker.

The reference is unbuildable in this image (C3 / cdyn / CBLAS are absent), so the oracle is pinned by
 (a) the known-answer tests the reference's own suite holds for this path
     (/root/reference/test/transition_prob/tprob_test.c -- restated here as inputs + expected
     outputs; the line ranges are cited per test), and
 (b) the outputs of the real reference recorded in SURVEY.md section 10.6
     (tests/golden/survey_known_answers.json).
CPU-only: runs under  pytest -m "not gpu".
"""
import json
import math
import os

import numpy as np
import pytest

from c3sc_amd import workloads as wl

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _linspace(lo, hi, n):
    i = np.arange(n, dtype=np.float64)
    return lo + (hi - lo) * i / float(n - 1)


# ----------------------------------------------------------------------------- (b) SURVEY 10.6
def test_survey_known_answer_transition_and_rhs(oracle):
    ka = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))
    t = ka["transition_assemble"]
    dx = t["dx"]
    h2 = t["hmin"] ** 2
    tv = []
    for hi in t["h"]:
        tv += [h2 / hi, h2 / hi / hi]
    diff = np.zeros((dx, dx))
    np.fill_diagonal(diff, t["sigma_diag"])
    res, prob, dt, _, _ = oracle.transition_assemble(dx, t["du"], t["dw"], h2, tv, t["drift"], diff.ravel())
    assert res == 0
    assert dt == pytest.approx(t["dt"], rel=1e-15)
    np.testing.assert_allclose(prob[:4], t["prob"], rtol=2e-15)
    assert abs(prob[4]) < 1e-15
    # the retired routine (nodeutil.c:82-233) gives the same no-gradient answer
    res2, prob2, dt2, _, _ = oracle.transition_assemble(dx, 1, dx, t["hmin"], t["h"], t["drift"], diff.ravel(), old=True)
    assert res2 == 0 and dt2 == dt
    np.testing.assert_array_equal(prob, prob2)

    b = ka["bellmanrhs"]
    val, _ = oracle.bellmanrhs(dx, 1, b["stage"], b["beta"], prob, dt, b["cost"])
    assert val == pytest.approx(b["value"], rel=1e-15)


def test_survey_known_answer_ft_vs_bruteforce(oracle):
    """reference valuef_eval_fiber_ind_nn == brute-force TT contraction (d=4,
    N={7,9,5,6}, ranks {1,3,4,2,1}, every dim_vary) to 1.4e-17; tprob_test.c:603-904 pins 1e-14."""
    N, ranks = [7, 9, 5, 6], [1, 3, 4, 2, 1]
    rng = np.random.default_rng(7)
    cores = [rng.uniform(-1, 1, size=(N[m], ranks[m] * ranks[m + 1])) for m in range(4)]
    vf = oracle.ValueF(N, ranks, cores)
    lb, ub = [-1.0] * 4, [1.0] * 4
    bnd = oracle.Boundary(lb, ub)
    for m in range(4):
        bnd.set_type(m, "reflect")
    xg = [_linspace(-1, 1, n) for n in N]
    worst = 0.0
    for k in range(4):
        fixed = [3, 4, 2, 1]
        x = np.array([[xg[m][j] if m == k else xg[m][fixed[m]] for m in range(4)] for j in range(N[k])])
        res, fi, dv = oracle.convert_fiber_to_ind(x, N, xg)
        assert res == 0 and dv == k
        _, ab, nv, nf = oracle.process_fibers_neighbor(fi, k, x, N, bnd)
        out = vf.eval_fiber_ind_nn(fi, k, nf, nv)
        for j in range(N[k]):
            ind = list(fixed)
            ind[k] = j
            assert abs(out[j, 8] - vf.eval_ind(ind)) <= 1e-14
            worst = max(worst, abs(out[j, 8] - vf.eval_ind(ind)))
            on = 0
            for m in range(4):
                for s in range(2):
                    nb = list(ind)
                    nb[m] = int(nv[2 * j + s]) if m == k else int(nf[on + s])
                    assert abs(out[j, 2 * m + s] - vf.eval_ind(nb)) <= 1e-14
                if m != k:
                    on += 2
    assert worst < 1e-14


# --------------------------------------------------- (a) tprob_test.c:327-364 Test_tprob_probsum
def _f1(pt, u):  # tprob_test.c:121-134
    return np.array([math.sin(pt[1] + pt[0]), pt[0] ** 2 * u[0]]), np.array([0.0, pt[0] ** 2])


def test_ref_tprob_probsum(oracle):
    dx, du, dw = 2, 1, 2
    h = [1e-1, 1e-2]
    drift, _ = _f1([-2.0, -0.3], [0.0])
    diff = np.eye(2).ravel()
    res, prob, dt, _, _ = oracle.transition_assemble(dx, du, dw, h[1], h, drift, diff, old=True)
    assert res == 0
    assert all(p > -1e-15 for p in prob)
    assert abs(prob.sum() - 1.0) <= 1e-15
    # new routine, same numbers
    tv = [h[1] ** 2 / h[0], h[1] ** 2 / h[0] / h[0], h[1] ** 2 / h[1], h[1] ** 2 / h[1] / h[1]]
    res, prob2, dt2, _, _ = oracle.transition_assemble(dx, du, dw, h[1] ** 2, tv, drift, diff)
    assert res == 0 and dt2 == pytest.approx(dt, rel=1e-15)
    np.testing.assert_allclose(prob2, prob, rtol=0, atol=1e-16)


# ------------------------------------------------------ tprob_test.c:366-432 Test_tprob_grad
@pytest.mark.parametrize("old", [True, False])
def test_ref_tprob_grad(oracle, old):
    rng = np.random.default_rng(11)
    dx, du, dw = 2, 1, 2
    h = [1e-1, 1e-2]
    hmin = h[1]
    tv = [hmin ** 2 / h[0], hmin ** 2 / h[0] ** 2, hmin ** 2 / h[1], hmin ** 2 / h[1] ** 2]
    args = (hmin, h) if old else (hmin ** 2, tv)
    diff = np.eye(2).ravel()
    gdiff = np.zeros(4)
    for _ in range(1000):
        pt = rng.uniform(-1.5, 1.5, 2)
        u = rng.uniform(-1.5, 1.5, 1)
        drift, gd = _f1(pt, u)
        res, prob, dt, gp, gdt = oracle.transition_assemble(dx, du, dw, *args, drift, diff, gd, gdiff, old=old)
        assert res == 0
        delta = 1e-10
        drift2, _ = _f1(pt, u + delta)
        res, prob2, dt2, _, _ = oracle.transition_assemble(dx, du, dw, *args, drift2, diff, old=old)
        assert res == 0
        np.testing.assert_allclose((prob2 - prob) / delta, gp, atol=1e-5)
        assert abs((dt2 - dt) / delta - gdt[0]) <= 1e-5


# ----------------------------------------------------- tprob_test.c:434-509 Test_tprob_grad2
def _f2(x, u):  # tprob_test.c:172-196 ; jac[i + j*dx]
    out = np.array([x[0] * x[2] ** 2 * u[0] * math.cos(u[1]), -x[1] * u[2], x[0] * x[1] * u[0] + 2 * u[1]])
    jac = np.array([x[0] * x[2] ** 2 * math.cos(u[1]), 0.0, x[0] * x[1],
                    x[0] * x[2] ** 2 * u[0] * (-math.sin(u[1])), 0.0, 2.0,
                    0.0, -x[1], 0.0])
    return out, jac


def test_ref_tprob_grad2(oracle):
    rng = np.random.default_rng(12)
    dx = du = dw = 3
    h = [1e-1, 1e-2, 1e0]
    hmin = h[1]
    diff = np.eye(3).ravel()
    gdiff = np.zeros(27)
    for _ in range(200):
        pt = rng.uniform(-1.5, 1.5, 3)
        u = rng.uniform(-1.5, 1.5, 3)
        drift, jac = _f2(pt, u)
        res, prob, dt, gp, gdt = oracle.transition_assemble(dx, du, dw, hmin, h, drift, diff, jac, gdiff, old=True)
        assert res == 0
        for i in range(du):
            delta = 1e-8
            u2 = u.copy()
            u2[i] += delta
            d2, _ = _f2(pt, u2)
            res, prob2, dt2, _, _ = oracle.transition_assemble(dx, du, dw, hmin, h, d2, diff, old=True)
            assert res == 0
            np.testing.assert_allclose((prob2 - prob) / delta, gp.reshape(2 * dx + 1, du)[:, i], atol=1e-4)
            assert abs((dt2 - dt) / delta - gdt[i]) <= 1e-5


def test_transition_dead_zone_and_stationary(oracle):
    """nodeutil.c:300-305 (|b| <= 1e-14 adds no drift), :365-367 (Q < 1e-14 returns 1, outputs untouched)."""
    dx = 2
    tv = [1e-3, 1e-2, 1e-2, 1.0]
    diff = np.diag([1.0, 0.5]).ravel()
    _, p0, dt0, _, _ = oracle.transition_assemble(dx, 1, dx, 1e-4, tv, [0.0, 0.0], diff)
    _, p1, dt1, _, _ = oracle.transition_assemble(dx, 1, dx, 1e-4, tv, [1e-14, -1e-14], diff)
    np.testing.assert_array_equal(p0, p1)
    assert dt0 == dt1
    _, p2, _, _, _ = oracle.transition_assemble(dx, 1, dx, 1e-4, tv, [1.0001e-14, 0.0], diff)
    assert p2[1] != p0[1]
    res, p3, dt3, _, _ = oracle.transition_assemble(dx, 1, dx, 1e-4, tv, [0.0, 0.0], np.zeros(4))
    assert res == 1 and np.isnan(dt3) and np.isnan(p3[4])  # dt, p_self untouched; p[0:4] hold the raw rates
    # ambiguous gradient -> 2 (nodeutil.c:346-349)
    res, *_ = oracle.transition_assemble(dx, 1, dx, 1e-4, tv, [0.0, 0.5], diff, np.zeros(2), np.zeros(4))
    assert res == 2


# ------------------------------------------------- tprob_test.c:1171-1249 Test_bellman_grad1
def test_ref_bellman_grad1(oracle):
    rng = np.random.default_rng(13)
    dx, du, dw = 2, 1, 2
    h = [1e-1, 1e-2]
    hmin = h[1]
    diff = np.eye(2).ravel()
    gdiff = np.zeros(4)
    discount = 0.1
    for _ in range(100):
        pt = rng.uniform(-1.5, 1.5, 2)
        u = rng.uniform(-1.5, 1.5, 1)
        cost = rng.uniform(0, 1, 5)
        stage = pt[0] ** 2 + pt[1] ** 2 + u[0] ** 2  # stagecost2d, tprob_test.c:239-258
        drift, gd = _f1(pt, u)
        res, prob, dt, gp, gdt = oracle.transition_assemble(dx, du, dw, hmin, h, drift, diff, gd, gdiff, old=True)
        val, grad = oracle.bellmanrhs(dx, du, stage, discount, prob, dt, cost, [2 * u[0]], gp, gdt)
        delta = 1e-9
        u2 = u + delta
        d2, _ = _f1(pt, u2)
        _, prob2, dt2, _, _ = oracle.transition_assemble(dx, du, dw, hmin, h, d2, diff, old=True)
        stage2 = pt[0] ** 2 + pt[1] ** 2 + u2[0] ** 2
        v2, _ = oracle.bellmanrhs(dx, du, stage2, discount, prob2, dt2, cost)
        assert abs((v2 - val) / delta - grad[0]) <= 1e-5


# ------------------------------------------------- tprob_test.c:1251-1378 Test_bellman_grad3d
@pytest.mark.parametrize("side", ["oracle", "libc3sc"])
def test_ref_bellman_grad3d(oracle, side):
    """The reference's 3-state / 3-control gradient test: drift f2 (:172-195), diffusion s2 = I (:197-220), stage cost
    stagecost3d (:273-300), h = (0.1, 0.01, 0.2), discount 0.1, 1000 random (point, control, neighbour-cost) draws plus the
    fixed first draw (:1291-1298); analytic gradients of the stage cost, every transition probability and of bellmanrhs
    against one-sided differences (delta 1e-7) to the reference's tolerance 1e-3 (relative once |fd| > 1, :1366-1370).
    Run on the oracle and on libc3sc.so's host functions (transition_assemble_old, bellmanrhs: pure host code)."""
    import ctypes as C

    dx = du = dw = 3
    h = [1e-1, 1e-2, 2e-1]
    hmin = h[1]
    diff, gdiff = np.eye(3).ravel(), np.zeros(27)
    discount = 0.1

    if side == "oracle":
        def assemble(drift, jac=None):
            res, prob, dt, gp, gdt = oracle.transition_assemble(dx, du, dw, hmin, h, drift, diff, jac, None if jac is None else gdiff, old=True)
            return res, prob, dt, gp, gdt

        def rhs(stage, prob, dt, cost, sg=None, gp=None, gdt=None):
            return oracle.bellmanrhs(dx, du, stage, discount, prob, dt, cost, sg, gp, gdt)
    else:
        import facade_lib as F

        L = F.lib()
        L.bellmanrhs.restype = C.c_double
        sz, cd = C.c_size_t, C.c_double

        def assemble(drift, jac=None):
            prob, dt = np.full(7, np.nan), C.c_double(np.nan)
            if jac is None:
                res = L.transition_assemble_old(sz(dx), sz(du), sz(dw), cd(hmin), F.dp(F.f64(h)), F.dp(F.f64(drift)), None,
                                                F.dp(diff), None, F.dp(prob), None, C.byref(dt), None, None)
                return res, prob, dt.value, None, None
            gp, gdt, space = np.zeros(21), np.zeros(3), np.zeros(3)
            res = L.transition_assemble_old(sz(dx), sz(du), sz(dw), cd(hmin), F.dp(F.f64(h)), F.dp(F.f64(drift)), F.dp(F.f64(jac)),
                                            F.dp(diff), F.dp(gdiff), F.dp(prob), F.dp(gp), C.byref(dt), F.dp(gdt), F.dp(space))
            return res, prob, dt.value, gp, gdt

        def rhs(stage, prob, dt, cost, sg=None, gp=None, gdt=None):
            if sg is None:
                return L.bellmanrhs(sz(dx), sz(du), cd(stage), None, cd(discount), F.dp(F.f64(prob)), None, cd(dt), None, F.dp(F.f64(cost)), None), None
            g = np.zeros(du)
            v = L.bellmanrhs(sz(dx), sz(du), cd(stage), F.dp(F.f64(sg)), cd(discount), F.dp(F.f64(prob)), F.dp(F.f64(gp)), cd(dt),
                             F.dp(F.f64(gdt)), F.dp(F.f64(cost)), F.dp(g))
            return v, g

    def stage3d(x, u):  # tprob_test.c:273-300
        v = 0.2 * x[0] * x[0] + 0.5 * x[1] * x[1] + 2.0 * x[2] * x[2] + 0.1 * u[0] * u[0] + 0.5 * u[1] * u[1] + 3.0 * u[2] * u[2]
        return v, np.array([0.2 * u[0], 1.0 * u[1], 6.0 * u[2]])

    rng = np.random.default_rng(14)
    delta = 1e-7
    for kk in range(1000):
        pt, u = rng.uniform(-1.5, 1.5, 3), rng.uniform(-1.5, 1.5, 3)
        if kk == 0:
            pt, u = np.array([-0.951613, -1.305556, -2.615385]), np.array([-5.0, -5.0, 0.0])
        cost = rng.uniform(0, 1, 7)
        stage, gstage = stage3d(pt, u)
        drift, jac = _f2(pt, u)
        res, prob, dt, gp, gdt = assemble(drift, jac)
        assert res != 1
        val, grad = rhs(stage, prob, dt, cost, gstage, gp, gdt)
        d0, _ = _f2(pt, u)
        _, prob3, dt3, _, _ = assemble(d0)
        new3, _ = rhs(stage, prob3, dt3, cost)
        assert new3 == val  # the value does not depend on whether gradients were requested
        for zz in range(du):
            v = u.copy()
            v[zz] += delta
            d2, _ = _f2(pt, v)
            res2, prob2, dt2, _, _ = assemble(d2)
            assert res2 == 0
            stage2, _ = stage3d(pt, v)
            new2, _ = rhs(stage2, prob2, dt2, cost)
            assert abs((stage2 - stage) / delta - gstage[zz]) <= 1e-3
            np.testing.assert_allclose((prob2 - prob3) / delta, gp.reshape(2 * dx + 1, du)[:, zz], rtol=0, atol=1e-3)
            fd = (new2 - new3) / delta
            err = abs(fd - grad[zz])
            if abs(fd) > 1:
                err /= abs(fd)
            assert err <= 1e-3


# ----------------------------------------- tprob_test.c:921-961 Test_valuef_fiber_to_ind
def test_ref_fiber_to_ind(oracle):
    N = [30, 43, 24]
    xg = [_linspace(-1.0, 2.0, n) for n in N]
    true = [10, 12, 13]
    for k in range(3):
        x = np.array([[xg[m][j] if m == k else xg[m][true[m]] for m in range(3)] for j in range(N[k])])
        res, fi, dv = oracle.convert_fiber_to_ind(x, N, xg)
        assert res == 0 and dv == k
        for m in range(3):
            if m != k:
                assert fi[m] == true[m]
        assert fi[k] == 0
    # error codes (nodeutil.c:433-435): off-grid -> 1 ; wrong N -> 2
    x = np.array([[xg[0][1] + 1e-9, xg[1][2], xg[2][3]]] * 2)
    assert oracle.convert_fiber_to_ind(x, N, xg)[0] == 1
    x = np.array([[xg[0][j], xg[1][2], xg[2][3]] for j in range(5)])
    assert oracle.convert_fiber_to_ind(x, N, xg)[0] == 2


# ------------------------------- tprob_test.c:1068-1169 Test_process_fibers_neighbor (exhaustive)
def test_ref_process_fibers_neighbor(oracle):
    lb, ub = [-1.0, -2.0, -3.0], [1.0, 2.0, 3.0]
    lengths = [0.8, 0.8, 0.8]
    bnd = oracle.Boundary(lb, ub)
    bnd.add_obstacle([0.0, 0.0, 0.0], lengths)
    N = [30, 43, 24]
    xg = [_linspace(lb[m], ub[m], N[m]) for m in range(3)]
    for aa in range(0, N[0], 1):
        for bb in range(0, N[1], 3):  # strided over the middle dim to keep the CPU suite short
            for cc in range(N[2]):
                fixed = [aa, bb, cc]
                for k in (1, 2):
                    x = np.array([[xg[m][j] if m == k else xg[m][fixed[m]] for m in range(3)] for j in range(N[k])])
                    truth = np.zeros(N[k], dtype=np.int32)
                    inside = (np.abs(x) < np.array(lengths) / 2.0).all(axis=1)
                    onb = ((x <= np.array(lb) + 1e-12) | (x >= np.array(ub) - 1e-12)).any(axis=1)
                    truth[onb] = 1
                    truth[inside] = -1
                    res, fi, dv = oracle.convert_fiber_to_ind(x, N, xg)
                    assert res == 0 and dv == k
                    res, ab, nv, nf = oracle.process_fibers_neighbor(fi, k, x, N, bnd)
                    assert res == 0
                    np.testing.assert_array_equal(ab, truth)


def test_process_fibers_neighbor_stencil_quirks(oracle):
    """nodeutil.c:515-612: periodic wrap (0 -> N-2, N-1 -> 1), reflect clamp, end-point overwrite (Q3)."""
    lb, ub = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]
    N = [5, 6, 7]
    xg = [_linspace(-1, 1, n) for n in N]
    bnd = oracle.Boundary(lb, ub)
    bnd.set_type(0, "periodic")
    bnd.set_type(1, "reflect")  # dim 2 stays absorbing
    bnd.add_obstacle([0.0, 0.0, 0.0], [0.2, 3.0, 3.0])  # covers x0 == 0 for every x1, x2

    def fiber(fixed, k):
        return np.array([[xg[m][j] if m == k else xg[m][fixed[m]] for m in range(3)] for j in range(N[k])])

    # vary dim 1 (reflect); dim 0 on the left periodic face, dim 2 interior
    _, ab, nv, nf = oracle.process_fibers_neighbor([0, 0, 3], 1, fiber([0, 0, 3], 1), N, bnd)
    assert list(nf) == [N[0] - 2, 1, 2, 4]
    assert list(nv[:2]) == [0, 1] and list(nv[-2:]) == [N[1] - 2, N[1] - 1]
    assert list(ab) == [0] * N[1]
    # dim 2 on the absorbing face -> whole fiber absorbed, but the reflect end points are reset to 0
    _, ab, nv, nf = oracle.process_fibers_neighbor([1, 0, 0], 1, fiber([1, 0, 0], 1), N, bnd)
    assert list(ab) == [0] + [1] * (N[1] - 2) + [0]
    assert list(nf) == [0, 2, 0, 0]
    assert list(nv[2:4]) == [1, 1]  # absorbed interior nodes point at themselves
    # obstacle fiber (x0 == 0) varying the periodic dim 0: only node 2 is inside; ends wrap
    _, ab, nv, nf = oracle.process_fibers_neighbor([0, 2, 3], 0, fiber([0, 2, 3], 0), N, bnd)
    assert list(ab) == [0, 0, -1, 0, 0]
    assert list(nv) == [3, 1, 0, 2, 2, 2, 2, 4, 3, 1]
    # obstacle marks at the end points are overwritten (to 1 for absorb): vary dim 2 through x0 == 0
    _, ab, _, _ = oracle.process_fibers_neighbor([2, 2, 0], 2, fiber([2, 2, 0], 2), N, bnd)
    assert list(ab) == [1] + [-1] * (N[2] - 2) + [1]


# -------------------------- tprob_test.c:535-919 Test_valuef_neighbor_eval (f = x0^2 + x0 x1 + x2^2)
def test_ref_valuef_neighbor_eval(oracle):
    """The reference builds the FT by cross approximation (C3, absent); f has an exact rank-3 nodal
    TT, built analytically here.  Pins: fiber routine == point evaluation at self and the 2d axis
    neighbours to 1e-14 for all three dim_vary with the test's index sets (tprob_test.c:575-577)."""
    N = [30, 43, 24]
    xg = [_linspace(-1.0, 2.0, n) for n in N]
    ranks = [1, 3, 3, 1]
    # row vector [x0^2, x0, 1] ; middle [[1,0,0],[x1,1? ...]] such that product = x0^2 + x0 x1 + x2^2
    G0 = np.zeros((N[0], 1, 3)); G0[:, 0, 0] = xg[0] ** 2; G0[:, 0, 1] = xg[0]; G0[:, 0, 2] = 1.0
    G1 = np.zeros((N[1], 3, 3)); G1[:, 0, 0] = 1.0; G1[:, 1, 0] = xg[1]; G1[:, 2, 1] = 1.0
    G2 = np.zeros((N[2], 3, 1)); G2[:, 0, 0] = 1.0; G2[:, 1, 0] = xg[2] ** 2
    cores = [np.ascontiguousarray(G.transpose(0, 2, 1)).reshape(G.shape[0], -1) for G in (G0, G1, G2)]
    vf = oracle.ValueF(N, ranks, cores)

    def f(i):
        return xg[0][i[0]] ** 2 + xg[0][i[0]] * xg[1][i[1]] + xg[2][i[2]] ** 2

    fixed = [3, 5, 9]
    for k in range(3):
        nb_fixed = [[4, 6, 8, 10], [1, 4, 4, 6], [2, 4, 4, 6]][k]
        Nk = N[k]
        nv = np.zeros(2 * Nk, dtype=np.uintp)
        nv[0], nv[1] = 0, 1
        for j in range(1, Nk):
            nv[2 * j], nv[2 * j + 1] = j - 1, min(j + 1, Nk - 1)
        out = vf.eval_fiber_ind_nn(fixed, k, nb_fixed, nv)
        for j in range(Nk):
            ind = list(fixed); ind[k] = j
            assert abs(out[j, 6] - f(ind)) <= 1e-14
            assert abs(out[j, 6] - vf.eval_ind(ind)) <= 1e-14
            on = 0
            for m in range(3):
                for s in range(2):
                    nb = list(ind)
                    nb[m] = int(nv[2 * j + s]) if m == k else nb_fixed[on + s]
                    assert abs(out[j, 2 * m + s] - f(nb)) <= 1e-14
                if m != k:
                    on += 2


# ------------------------------------------------------------------ hashgrid.c:49-87 (bit-exact)
def test_key_and_hash_bit_exact(oracle):
    assert oracle.key_string([3, 14, 0, 159, 0, 2]) == b"3 14 0 159 0 2 "
    assert oracle.key_string([0]) == b"0 "
    assert oracle.key_string([40, 40, 40, 40, 40, 40, 40, 0, 12]) == b"40 40 40 40 40 40 40 0 12 "

    def h(s):
        v = 0
        for ch in s:
            v = (ch + (v << 5) - v) & 0xFFFFFFFFFFFFFFFF
        return v % 1000000

    for key in (b"0 ", b"3 14 0 159 0 2 ", b"40 40 40 40 40 40 40 0 12 ", b"100 99 98 0 1000 "):
        assert oracle.hashchar(1000000, key) == h(key)
    assert oracle.hashchar(1000000, b"1 2 3 ") == 902986  # independent Python evaluation of h = c + 31*h (mod 2^64) mod 1e6


# ------------------------------------------- bellman.c:171-188, :1962-1999 grid constants
def test_grid_refs_and_problem_grids(oracle):
    w = wl.c4_car7d(n=11, r=3)
    P = oracle.Problem(w)
    xg = w.xgrid()
    hs = []
    for m in range(w.dx):
        g = P.xgrid(m)
        np.testing.assert_array_equal(g, xg[m])
        hs.append(g[1] - g[0])
    hmin = min(hs + [w.ub[0] - w.lb[0]])
    assert P.h2() == hmin * hmin
    t = P.tvec()
    for m in range(w.dx):
        assert t[2 * m] == hmin * hmin / hs[m]
        assert t[2 * m + 1] == hmin * hmin / hs[m] / hs[m]


def test_rossler_model_restates_the_reference_example(oracle):
    """examples/rossler/rossler.c:80-157 (a = b = 0.1, c = 14; the control enters the second equation; diffusion diag
    (s, s, s_last); stage 100 |x|^2 + u^2; boundary cost 1000; obstacle cost 0), evaluated here from those formulas."""
    import ctypes as C

    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    prm = (C.c_double * 8)(3.0, 0.7, 1.3)
    dx, du = C.c_size_t(), C.c_size_t()
    assert L.orc_model_dims(wl.MODEL_ROSSLER3D, prm, C.byref(dx), C.byref(du)) == 0 and (dx.value, du.value) == (3, 1)
    rng = np.random.default_rng(5)
    for _ in range(20):
        x = rng.uniform(-1.0, 1.0, 3); u = rng.uniform(-1.0, 1.0, 1)
        out = np.zeros(3); s = C.c_double()
        px, pu, po = x.ctypes.data_as(dp), u.ctypes.data_as(dp), out.ctypes.data_as(dp)
        assert L.orc_model_drift(wl.MODEL_ROSSLER3D, prm, px, pu, po) == 0
        np.testing.assert_array_equal(out, [-x[1] - x[2], x[0] + 0.1 * x[1] + u[0], 0.1 + x[2] * (x[0] - 14.0)])
        assert L.orc_model_diff_diag(wl.MODEL_ROSSLER3D, prm, px, pu, po) == 0
        np.testing.assert_array_equal(out, [0.7, 0.7, 1.3])
        assert L.orc_model_stage(wl.MODEL_ROSSLER3D, prm, px, pu, C.byref(s)) == 0
        assert s.value == ((0.0 + 1e2 * x[0] * x[0]) + 1e2 * x[1] * x[1]) + 1e2 * x[2] * x[2] + 1.0 * u[0] * u[0]
        assert L.orc_model_boundcost(wl.MODEL_ROSSLER3D, prm, px, C.byref(s)) == 0 and s.value == 1000.0
        assert L.orc_model_obscost(wl.MODEL_ROSSLER3D, prm, px, C.byref(s)) == 0 and s.value == 0.0
    w = wl.WORKLOADS["rossler3d"]()
    assert w.bc == (wl.BC_REFLECT,) * 3 and w.discount == 0.1 and w.lb == (-1.0,) * 3 and w.ub == (1.0,) * 3  # rossler.c:208-307
    assert w.ngrid == (20,) * 3 and w.cands.min() == -4.0 and w.cands.max() == 4.0


# ------------------------------------------------- not in the reference: the consistent end-point rule of the solver loops
def test_consistent_ends_make_the_fiber_function_a_function_of_the_node(oracle):
    """process_fibers_neighbor resets the flags of a fiber's end points from the varying dimension's own boundary type
    (nodeutil.c:570-612): on a problem with absorbing AND reflecting / periodic dimensions the value of a node
    on an absorbing face then depends on the direction of the fiber it is computed in -- the literal oracle shows it (that is
    the reference's behaviour), and with orc_boundary_set_consistent_ends the same backup gives ONE value per node whatever
    the direction.  The consistent value is always one of the literal ones (the one every non-end-point direction gives),
    and nodes off the absorbing faces / obstacle end points are untouched."""
    import itertools

    from c3sc_amd import workloads as wl

    w = wl.c2_dubins().scaled(ngrid=(7, 6, 9), rank=3)
    cores = wl.smooth_cores(w)
    lit, con = oracle.Problem(w, cores), oracle.Problem(w, cores, consistent_ends=True)
    dense = {}
    for name, P in (("lit", lit), ("con", con)):
        for k in range(3):
            dims = [range(n) if m != k else [0] for m, n in enumerate(w.ngrid)]
            idx = np.array(list(itertools.product(*dims)), dtype=np.int32)
            out, _, ab = P.bellman_fibers(k, idx)
            shp = [n for m, n in enumerate(w.ngrid) if m != k] + [w.ngrid[k]]
            dense[name, k] = (np.moveaxis(out.reshape(shp), -1, k), np.moveaxis(ab.reshape(shp), -1, k))
    # literal: x / y absorbing, theta periodic -> direction 2 disagrees with direction 0 at theta's end points on the x / y faces
    assert np.abs(dense["lit", 2][0] - dense["lit", 0][0]).max() > 1.0
    for k in (1, 2):
        np.testing.assert_array_equal(dense["con", k][1], dense["con", 0][1])                       # flags: bit-exact
        assert np.abs(dense["con", k][0] - dense["con", 0][0]).max() <= 1e-13 * np.abs(dense["con", 0][0]).max()
    # the consistent tensor equals the literal one of a direction whose end points are absorbing anyway (x: dim 0)
    np.testing.assert_array_equal(dense["con", 0][1], dense["lit", 0][1])
    assert np.abs(dense["con", 0][0] - dense["lit", 0][0]).max() == 0.0
    # and along theta it differs from the literal rule at end points only
    d = dense["con", 2][0] != dense["lit", 2][0]
    assert d.any() and not d[:, :, 1:-1].any()


def test_tprob3d_model_restates_the_reference_tests_callbacks(oracle):
    """ORC_MODEL_TPROB3D against the formulas of test/transition_prob/tprob_test.c (f3 :223-251, s2 :197-220, stagecost3d
    :273-300, boundcost :302-309, ocost :311-318) written out here."""
    import ctypes as C

    from c3sc_amd import workloads as wl

    L = oracle.lib()
    rng = np.random.default_rng(21)
    dx, du = C.c_size_t(0), C.c_size_t(0)
    assert L.orc_model_dims(wl.MODEL_TPROB3D, None, C.byref(dx), C.byref(du)) == 0 and (dx.value, du.value) == (3, 3)
    for _ in range(50):
        x, u = rng.uniform(-3, 3, 3), rng.uniform(-5, 5, 3)
        out = np.zeros(3)
        px, pu, po = oracle.dp(x), oracle.dp(u), oracle.dp(out)
        assert L.orc_model_drift(wl.MODEL_TPROB3D, None, px, pu, po) == 0
        np.testing.assert_array_equal(out, [x[0] * x[2] ** 2 * u[0], -x[1] * u[2] + u[1], x[0] * x[1] * u[0] + 2 * u[1]])
        assert L.orc_model_diff_diag(wl.MODEL_TPROB3D, None, px, pu, po) == 0
        np.testing.assert_array_equal(out, [1.0, 1.0, 1.0])
        s = C.c_double(0)
        assert L.orc_model_stage(wl.MODEL_TPROB3D, None, px, pu, C.byref(s)) == 0
        want = 0.0
        for t in (0.2 * x[0] * x[0], 0.5 * x[1] * x[1], 2.0 * x[2] * x[2], 0.1 * u[0] * u[0], 0.5 * u[1] * u[1], 3.0 * u[2] * u[2]):
            want += t
        assert s.value == want
        assert L.orc_model_boundcost(wl.MODEL_TPROB3D, None, px, C.byref(s)) == 0 and s.value == 100.0
        assert L.orc_model_obscost(wl.MODEL_TPROB3D, None, px, C.byref(s)) == 0 and s.value == 0.0


def test_perch7d_model_restates_the_example(oracle):
    """ORC_MODEL_PERCH7D against examples/perching/perch.c:36-241 written out here with Python's libm (the oracle keeps the
    reference's own expressions: cos / sin / atan2); and the identity the device functor uses instead --
    |v|^2 sin(a - atan2(v_z, v_x)) == |v| (sin a v_x - cos a v_z) -- to 1e-13 relative of the force scale."""
    import ctypes as C
    import math

    from c3sc_amd import workloads as wl

    L = oracle.lib()
    w = wl.WORKLOADS["perch7d"]()
    rng = np.random.default_rng(22)
    m, g, rho, S_w, S_e, In, l, l_w, l_e = 0.05, 9.81, 1.292, 0.1, 0.025, 6e-3, 0.35, -0.03, 0.04
    for _ in range(200):
        x = rng.uniform(w.lb, w.ub)
        u = rng.uniform(-2 * math.pi, 2 * math.pi, 1)
        c_t, s_t, c_tp, s_tp, c_p = math.cos(x[2]), math.sin(x[2]), math.cos(x[2] + x[3]), math.sin(x[2] + x[3]), math.cos(x[3])
        dw = (x[4] + l_w * x[6] * s_t, x[5] - l_w * x[6] * c_t)
        de = (x[4] + l * x[6] * s_t + l_e * (x[6] + u[0]) * s_tp, x[5] - l * x[6] * c_t - l_e * (x[6] + u[0]) * c_tp)
        f_w = rho * S_w * (dw[0] * dw[0] + dw[1] * dw[1]) * math.sin(x[2] - math.atan2(dw[1], dw[0]))
        f_e = rho * S_e * (de[0] * de[0] + de[1] * de[1]) * math.sin(x[2] + x[3] - math.atan2(de[1], de[0]))
        want = [x[4], x[5], x[6], u[0], (-f_w * s_t - f_e * s_tp) / m, (f_w * c_t + f_e * c_tp - m * g) / m, (-f_w * l_w - f_e * (l * c_p + l_e)) / In]
        out = np.zeros(7)
        assert L.orc_model_drift(wl.MODEL_PERCH7D, None, oracle.dp(x), oracle.dp(u), oracle.dp(out)) == 0
        np.testing.assert_allclose(out, want, rtol=4e-16, atol=0)  # the compiler may associate a product differently: an ulp
        f_w2 = rho * S_w * math.hypot(*dw) * (s_t * dw[0] - c_t * dw[1])
        f_e2 = rho * S_e * math.hypot(*de) * (s_tp * de[0] - c_tp * de[1])
        assert abs(f_w2 - f_w) <= 1e-13 * (1.0 + abs(f_w)) and abs(f_e2 - f_e) <= 1e-13 * (1.0 + abs(f_e))
        s = C.c_double(0)
        assert L.orc_model_stage(wl.MODEL_PERCH7D, None, oracle.dp(x), oracle.dp(u), C.byref(s)) == 0
        want_s = 0.0
        for t in (20.0 * x[0] * x[0], 50.0 * x[1] * x[1], 10.0 * x[2] * x[2], 1.0 * x[3] * x[3], 1.0 * x[4] * x[4], 1.0 * x[5] * x[5], 1.0 * x[6] * x[6],
                  0.1 * u[0] * u[0]):
            want_s += t
        assert s.value == want_s
        assert L.orc_model_boundcost(wl.MODEL_PERCH7D, None, oracle.dp(x), C.byref(s)) == 0
        want_b = 0.0
        for t in (600.0 * x[0] * x[0], 400.0 * x[1] * x[1], 1.0 / 9.0 * x[2] * x[2], 5.0 * (x[2] - math.pi / 2.0) * (x[2] - math.pi / 2.0),
                  1.0 / 9.0 * x[3] * x[3], 1.0 * x[4] * x[4], 1.0 * (x[5] + 1.5) * (x[5] + 1.5), 1.0 / 9.0 * (x[6] + 0.5) * (x[6] + 0.5)):
            want_b += t
        assert s.value == want_b
        out7 = np.zeros(7)
        assert L.orc_model_diff_diag(wl.MODEL_PERCH7D, None, oracle.dp(x), oracle.dp(u), oracle.dp(out7)) == 0 and (out7 == 1e-9).all()


def test_skid5d_and_cothrust6d_models_restate_the_examples(oracle):
    """ORC_MODEL_SKID5D against examples/skidding5d/scar.c:39-163 and ORC_MODEL_COTHRUST6D against
    examples/cothrust2/copterposethrust.c:40-209, the callbacks written out here with Python's libm.  skidding5d's diffusion
    callback writes out[28] of its 25-element matrix where out[18] was meant: the yaw rate's diagonal entry is
    therefore 0 in the reference, and must be 0 here -- mirrored, not fixed."""
    import ctypes as C
    import math

    from c3sc_amd import workloads as wl

    L = oracle.lib()
    rng = np.random.default_rng(31)
    w = wl.WORKLOADS["skid5d"]()
    assert w.bc == (wl.BC_REFLECT, wl.BC_REFLECT, wl.BC_PERIODIC, wl.BC_ABSORB, wl.BC_ABSORB) and w.discount == 1.0 and w.ncand == 20
    m, cf, ct, a, b, In, s = 1460.0, 17000.0, 20000.0, 1.2, 1.5, 2170.0, 27.0
    for _ in range(200):
        x = rng.uniform(w.lb, w.ub)
        u = rng.uniform(-5.0 * math.pi / 180.0, 5.0 * math.pi / 180.0, 1)
        co, so = math.cos(x[2]), math.sin(x[2])
        ff = cf * ((x[4] + a * x[3]) / s + u[0])
        ft = ct * (x[4] - b * x[3]) / s
        want = [s * co - x[4] * so, s * so + x[4] * co, x[3], (a * ff - b * ft) / In, -s * x[3] + (ff + ft) / m]
        out = np.zeros(5)
        assert L.orc_model_drift(wl.MODEL_SKID5D, None, oracle.dp(x), oracle.dp(u), oracle.dp(out)) == 0
        np.testing.assert_allclose(out, want, rtol=4e-16, atol=0)
        sv = C.c_double(0)
        assert L.orc_model_stage(wl.MODEL_SKID5D, None, oracle.dp(x), oracle.dp(u), C.byref(sv)) == 0
        o = 1.0 + 0.02 * x[0] ** 2 + 0.02 * x[1] ** 2
        o = o + x[3] ** 2 + x[4] ** 2
        assert sv.value == pytest.approx(o, rel=4e-16)
        assert L.orc_model_boundcost(wl.MODEL_SKID5D, None, oracle.dp(x), C.byref(sv)) == 0
        o = 0.1 * x[0] ** 2 + 0.1 * x[1] ** 2
        o = o + 0.1 * x[3] ** 2 + 0.1 * x[4] ** 2
        assert sv.value == pytest.approx(o, rel=4e-16)
        sg = np.full(5, np.nan)
        assert L.orc_model_diff_diag(wl.MODEL_SKID5D, None, oracle.dp(x), oracle.dp(u), oracle.dp(sg)) == 0
        assert list(sg) == [1e-5, 1e-5, 1e-5, 0.0, 1e-5]  # Q13: out[28] instead of out[18]
    w = wl.WORKLOADS["cothrust6d"]()
    assert w.bc == (wl.BC_REFLECT,) * 6 and w.discount == 1.0 and w.ncand == 64
    mq, g = 1.227, 9.81
    mg = mq * g
    for _ in range(200):
        x = rng.uniform(w.lb, w.ub)
        u = rng.uniform([-1.5, -0.4, -0.4], [1.5, 0.4, 0.4])
        cphi, sphi, cth, sth = math.cos(u[1]), math.sin(u[1]), math.cos(u[2]), math.sin(u[2])
        want = [x[3], x[4], x[5], cphi * sth * (u[0] - mg) / mq, -sphi * (u[0] - mg) / mq, g + cth * cphi * (u[0] - mg) / mq]
        out = np.zeros(6)
        assert L.orc_model_drift(wl.MODEL_COTHRUST6D, None, oracle.dp(x), oracle.dp(u), oracle.dp(out)) == 0
        np.testing.assert_allclose(out, want, rtol=4e-16, atol=0)
        sv = C.c_double(0)
        assert L.orc_model_stage(wl.MODEL_COTHRUST6D, None, oracle.dp(x), oracle.dp(u), C.byref(sv)) == 0
        o = 0.0
        o = o + 60.0 + 2 * u[0] ** 2 + 1 * u[1] ** 2 + 6 * u[2] ** 2
        o = o + 8.0 * x[2] ** 2
        o = o + 6.0 * x[1] ** 2
        o = o + 8.0 * x[0] ** 2
        assert sv.value == pytest.approx(o, rel=4e-16)
        assert L.orc_model_boundcost(wl.MODEL_COTHRUST6D, None, oracle.dp(x), C.byref(sv)) == 0 and sv.value == 10.0
        sg = np.zeros(6)
        assert L.orc_model_diff_diag(wl.MODEL_COTHRUST6D, None, oracle.dp(x), oracle.dp(u), oracle.dp(sg)) == 0
        assert list(sg) == [1e-1, 1e-1, 2e-1, 12e-1, 12e-1, 12e-1]
    # the callback refuses controls outside its box (copterposethrust.c:92-100)
    assert L.orc_model_drift(wl.MODEL_COTHRUST6D, None, oracle.dp(np.zeros(6)), oracle.dp(np.array([2.0, 0.0, 0.0])), oracle.dp(np.zeros(6))) == 1
