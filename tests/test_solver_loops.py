"""Value / policy iteration loops through the reference's API names (c3control_init_value, step_vi, vi_solve,
pi_solve) over the own cross driver, with every core step of the interpolation running as one batched kernel
launch.  Pinned against the CPU path: the same driver fed by the oracle's bellman_vi (SURVEY.md 8d metric iii:
nodal L-inf error after a fixed number of iterations, target <= 1e-6)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from c3sc_amd import workloads as wl  # noqa: E402

pytestmark = pytest.mark.gpu
FIBER_FN = C.CFUNCTYPE(C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def _setup(w, maxrank=12, startrank=3, round_tol=1e-9, kick=3, crossrank=0, cross_maxiter=0):
    import facade_lib

    L = facade_lib.lib()
    for n in ("c3control_init_value", "c3control_step_vi", "c3control_vi_solve", "c3control_pi_solve", "valuef_interp"):
        getattr(L, n).restype = C.c_void_p
    for n in ("valuef_norm", "valuef_norm2diff", "valuef_eval_ind", "diag_last_diff"):
        getattr(L, n).restype = C.c_double
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
    L.diag_count.restype = C.c_size_t
    ctl = facade_lib.Control(w)
    aa = C.c_void_p(L.approx_args_init())
    L.approx_args_set_cross_tol(aa, C.c_double(1e-10))
    L.approx_args_set_round_tol(aa, C.c_double(round_tol))
    L.approx_args_set_kickrank(aa, C.c_size_t(kick))
    L.approx_args_set_startrank(aa, C.c_size_t(startrank))
    L.approx_args_set_maxrank(aa, C.c_size_t(maxrank))
    if crossrank:
        L.approx_args_set_crossrank(aa, C.c_size_t(crossrank))  # the cross approximation runs above maxrank, its result is rounded to it
    if cross_maxiter:
        L.approx_args_set_cross_maxiter(aa, C.c_size_t(cross_maxiter))  # cap on the cross iterations per interpolation (the reference's: 5)
    return L, facade_lib, ctl, aa


def _cores_of(L, vf, w):
    ranks = [L.valuef_get_ranks(vf)[i] for i in range(w.dx + 1)]
    pp = L.valuef_get_cores(vf)
    cores = [np.ctypeslib.as_array(pp[m], shape=(w.ngrid[m] * ranks[m] * ranks[m + 1],)).copy() for m in range(w.dx)]
    return ranks, cores


def _all_values(L, fl, vf, w):
    out = np.zeros(w.ngrid)
    for ind in np.ndindex(*w.ngrid):
        out[ind] = L.valuef_eval_ind(vf, fl.sp(np.array(ind, dtype=np.uintp)))
    return out


def test_value_iteration_gpu_path_matches_cpu_path(oracle):
    # maxrank >= min N: rank adaptation runs until rounding (1e-9) drops a rank, so the two paths agree to the
    # truncation error whatever pivots they pick (1e-16 differences in the fiber values can flip pivot choices; the
    # driver's confirmation round keeps such a flip from ending the adaptation early, see
    # test_cross_driver.py::test_rank_adaptation_survives_degenerate_index_sets)
    w = wl.c1_lqg2d().scaled(ngrid=(19, 17))
    L, fl, ctl, aa = _setup(w, maxrank=17)
    const = FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).fill(0.2), 0)[1])
    v_gpu = C.c_void_p(L.c3control_init_value(ctl.h, const, None, aa, 0))
    assert L.valuef_norm(v_gpu) == pytest.approx(0.8, rel=1e-12)  # constant 0.2 on [-2,2]^2
    # CPU path: the same cross driver, fibers evaluated by the oracle's bellman_vi (with the reference's memo)
    xg = ctl.xgrid()
    gs = [fl.f64(g) for g in xg]
    gp = fl.ptrs(gs)
    Ng = np.array(w.ngrid, dtype=np.uintp)
    v_cpu = C.c_void_p(L.valuef_copy(v_gpu))
    state = {}

    def cpu_fiber(n, x, out, a):
        X = np.ctypeslib.as_array(x, shape=(n, w.dx)).copy()
        np.ctypeslib.as_array(out, shape=(n,))[:] = state["P"].bellman_vi(X, use_memo=True)[0]
        return 0

    cpu_cb = FIBER_FN(cpu_fiber)
    ne = C.c_size_t(0)
    for it in range(4):
        nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v_gpu, aa, ctl.opt, 0, C.byref(ne)))
        L.valuef_destroy(v_gpu)
        v_gpu = nxt
        ranks, cores = _cores_of(L, v_cpu, w)
        wr = wl.Workload(w.name, w.model, w.params, w.dx, w.du, w.lb, w.ub, w.ngrid, tuple(ranks), w.discount, w.bc,
                         list(w.obstacles), w.cands)
        P = oracle.Problem(wr, [c.reshape(w.ngrid[m], -1) for m, c in enumerate(cores)])
        P.increment_vi_iter()
        state["P"] = P
        nxt = C.c_void_p(L.valuef_interp(C.c_size_t(w.dx), cpu_cb, None, fl.sp(Ng), gp, v_cpu, aa, 0))
        L.valuef_destroy(v_cpu)
        v_cpu = nxt
        assert ne.value > 0
    a, b = _all_values(L, fl, v_gpu, w), _all_values(L, fl, v_cpu, w)
    print("gpu-vs-cpu path, max nodal difference after 4 sweeps:", np.abs(a - b).max(), "scale", np.abs(b).max())
    assert np.abs(a - b).max() <= 1e-7 * max(1.0, np.abs(b).max())  # SURVEY 8d (iii) asks for 1e-6
    assert L.valuef_norm2diff(v_gpu, v_cpu) <= 1e-7 * L.valuef_norm(v_cpu)
    L.valuef_destroy(v_gpu)
    L.valuef_destroy(v_cpu)
    L.approx_args_free(aa)
    ctl.close()


def test_car7d_outer_loop_device_vs_oracle_side_by_side(oracle):
    """The examples' outer loop (pi_solve(10) + one vi_solve step per control update, e.g. dubinscar.c:343-352) on a reduced
    7-D car grid, 20 control updates: each update run on the device path and on the oracle-fed path from the SAME state
    is compared node by node -- every one of them must agree to 1e-6 of max |V| (north_star; measured 3e-14).
    Round 2 had to tolerate a quarter of the updates disagreeing by 15-27 %.  The cause was not the rank cap: with the
    reference's literal end-point rule (nodeutil.c:570-612) a node on an absorbing face has one value as the
    end point of a reflecting / periodic fiber and another along every other direction, the memo keeps whichever came first,
    and the function handed to the cross approximation is then inconsistent at a few per cent of the nodes by the size of the
    boundary cost -- its rank-5 cross approximation was off by 88 % where 0.4 % is attainable.  The solver
    loops now keep the end points' flags (c3control_set_consistent_ends, mirrored in the oracle); "literal_ends" in the
    configuration restores the reference's rule.
    The data avoid EXACT ties between candidates (the tie-break of the brute-force scan lives in C3): with the
    symmetric 3 x 3 candidate grid and a start value that does not depend on the steering / acceleration states, +u and -u tie
    at every node -- so the candidate list is slightly asymmetric and the start value depends on every coordinate."""
    import regression_lib as R

    w0 = wl.c4_car7d().scaled(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4)
    cands = np.array([[a, b] for a in (-0.5, 0.07, 0.43) for b in (-1.0, 0.13, 0.91)])
    w = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, w0.ranks, w0.discount, w0.bc, list(w0.obstacles), cands)
    wts = np.array([0.3, 0.5, 0.2, 0.1, 0.15, 0.7, 0.25])
    cfg = dict(w=w, max_updates=21, conv=1e-9, adapt=1, startrank=3, maxrank=5, kick=2, cross_tol=1e-10, round_tol=1e-9,
               start_fn=lambda X: 1.0 + ((X - 0.1) ** 2 * wts).sum(axis=1))
    gpu, orc = R.GpuLoop(cfg), R.OracleLoop(cfg)
    Lb = gpu.L
    state = gpu.run(max_updates=1)
    worst, diffs = 0.0, []
    for _ in range(20):
        a = gpu.run(max_updates=1, cost=C.c_void_p(Lb.valuef_copy(state)))
        b = orc.run(max_updates=1, cost=C.c_void_p(Lb.valuef_copy(state)))
        vb = orc.nodal(b)
        diffs.append(np.abs(gpu.nodal(a) - vb).max() / np.abs(vb).max())
        worst = max(worst, diffs[-1])
        Lb.valuef_destroy(b)
        Lb.valuef_destroy(state)
        state = a
    print("per update:", " ".join(f"{x:.1e}" for x in diffs))
    print(f"car7d {w.ngrid}: 20 control updates in lock-step: worst {worst:.1e}, median {np.median(diffs):.1e}; |V| = {gpu.norm(state):.6e}, rank {gpu.rank(state)}")
    assert worst <= 1e-6
    Lb.valuef_destroy(state)
    gpu.close()
    orc.close()


def test_literal_end_point_rule_outer_loop_device_vs_oracle_side_by_side(oracle):
    """The REFERENCE'S OWN end-point rule (process_fibers_neighbor, nodeutil.c:570-612: the flags of a fiber's two end points come
    from the varying dimension's boundary type, whatever the fixed dimensions say -- C3SC_LITERAL_ENDS=1 /
    c3control_set_consistent_ends(0)) at solver level: the Dubins car has two absorbing dimensions, a periodic one and an obstacle,
    so nodes on an absorbing face that are end points of a periodic fiber, and obstacle nodes that are end points, take the value of
    whichever fiber reaches them first (memo: first stored wins, bellman.c:1349-1353).  The device-resident loop (device memo, same
    order of core steps) and the oracle-fed loop (string memo) must make the same choice at every such node: every control update
    from the same state agrees to 1e-6 of max |V| node by node.  The solver's default (consistent end points) is covered by the car7d
    test above; this one keeps the literal rule covered at solver level."""
    import regression_lib as R

    w0 = wl.c2_dubins().scaled(ngrid=(21, 19, 17), rank=4)
    w = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, w0.ranks, w0.discount, w0.bc, list(w0.obstacles),
                    np.array([[-1.0], [0.07], [0.93]]))  # no exact ties between +u and -u (the scan's tie-break lives in C3)
    cfg = dict(w=w, max_updates=13, conv=1e-9, adapt=1, startrank=4, maxrank=17, kick=3, cross_tol=1e-10, round_tol=1e-9, literal_ends=True,
               start_fn=lambda X: 1.0 + ((X - 0.1) ** 2 * np.array([0.3, 0.5, 0.2])).sum(axis=1))
    gpu, orc = R.GpuLoop(cfg), R.OracleLoop(cfg)
    assert not gpu.consistent_ends and not orc.consistent_ends
    Lb = gpu.L
    # the rule really is in force: the literal and the consistent backup of the start value differ at end points (oracle, one fiber batch)
    state = gpu.run(max_updates=1)
    worst, diffs = 0.0, []
    for _ in range(12):
        a = gpu.run(max_updates=1, cost=C.c_void_p(Lb.valuef_copy(state)))
        b = orc.run(max_updates=1, cost=C.c_void_p(Lb.valuef_copy(state)))
        vb = orc.nodal(b)
        diffs.append(np.abs(gpu.nodal(a) - vb).max() / np.abs(vb).max())
        worst = max(worst, diffs[-1])
        Lb.valuef_destroy(b)
        Lb.valuef_destroy(state)
        state = a
    print("per update (literal end points):", " ".join(f"{x:.1e}" for x in diffs))
    print(f"dubins3d {w.ngrid}, literal end-point rule: 12 control updates in lock-step: worst {worst:.1e}; |V| = {gpu.norm(state):.6e}, rank {gpu.rank(state)}")
    assert worst <= 1e-6
    Lb.valuef_destroy(state)
    gpu.close()
    orc.close()


def test_rossler_example_outer_loop_device_vs_oracle_side_by_side(oracle):
    """examples/rossler/rossler.c:208-360 with its own settings (N = 20 on [-1,1]^3, reflecting box, beta = 0.1, start cost
    10 |x|^2, rank adaptation 2 -> N with kick 2, cross tol 1e-10, rounding 1e-8, pi_solve(10) + one vi step per update) and
    a candidate list over its control box [-4, 4] (slightly asymmetric: no exact +u / -u ties, see the car7d test): ten
    control updates, each run on the device path and on the oracle-fed path from the same state."""
    import regression_lib as R

    w0 = wl.WORKLOADS["rossler3d"]()
    cands = (np.linspace(-4.0, 4.0, 33) + 0.013).clip(-4.0, 4.0).reshape(-1, 1)
    w = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, w0.ranks, w0.discount, w0.bc, [], cands)
    cfg = dict(w=w, max_updates=11, conv=1e-8, adapt=1, startrank=2, maxrank=20, kick=2, cross_tol=1e-10, round_tol=1e-8,
               start_fn=lambda X: 10.0 * (X ** 2).sum(axis=1))
    gpu, orc = R.GpuLoop(cfg), R.OracleLoop(cfg)
    Lb = gpu.L
    state = gpu.run(max_updates=1)
    diffs = []
    for _ in range(10):
        a = gpu.run(max_updates=1, cost=C.c_void_p(Lb.valuef_copy(state)))
        b = orc.run(max_updates=1, cost=C.c_void_p(Lb.valuef_copy(state)))
        vb = orc.nodal(b)
        diffs.append(np.abs(gpu.nodal(a) - vb).max() / np.abs(vb).max())
        Lb.valuef_destroy(b)
        Lb.valuef_destroy(state)
        state = a
    print("per update:", " ".join(f"{x:.1e}" for x in diffs))
    print(f"rossler3d {w.ngrid}: 10 control updates in lock-step, worst {max(diffs):.1e}; |V| = {gpu.norm(state):.6e}, rank {gpu.rank(state)}")
    assert max(diffs) <= 1e-8  # the rounding tolerance of the example; typical agreement is ~1e-13
    Lb.valuef_destroy(state)
    gpu.close()
    orc.close()


@pytest.mark.parametrize("name,kw,aargs,must_take", [
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=2), False),
    ("dubins3d", dict(ngrid=(41, 41, 41), rank=4), dict(maxrank=12, kick=3), True),
    ("lqg2d", dict(ngrid=(60, 60), rank=4), dict(maxrank=12, kick=3), True),
], ids=["car7d-small", "dubins3d", "lqg2d"])
def test_speculative_first_iteration_same_bits_and_actually_used(name, kw, aargs, must_take):
    """A sweep whose predecessor ended with the index sets it started from first tries the whole iteration in d + 1 launches
    (c3sc_hip_cross_speculate: the fiber lists of all d cores from the current sets back to back, bellman_vi's or bellman_pi's,
    then every core step side by side).  When every step reproduces its set that WAS the iteration; otherwise the sequential one
    runs.  Either way cores, ranks, sweep counts and last steps of the examples' control updates (policy-evaluation sweeps + one
    value-iteration sweep) must be the bits of the sequential device iteration (C3SC_NO_SPECULATE=1) and of the host-driven driver
    (C3SC_HOST_CROSS=1) -- and on the converging problems the short cut must actually have been taken.  The short cut exists
    only under the solver loops' own end-point rule (c3control_set_consistent_ends, the library default; this file's other tests
    run the reference's literal rule): with it a node's value does not depend on the fiber that computes it."""
    w = wl.WORKLOADS[name]().scaled(**kw)
    L, fl, ctl, aa = _setup(w, **aargs)
    L.c3control_set_consistent_ends(ctl.h, C.c_int(1))
    d = w.dx

    def start(n, x, out, a):
        X = np.ctypeslib.as_array(x, shape=(n, d))
        np.ctypeslib.as_array(out, shape=(n,))[:] = 1.0 + 0.1 * ((X - 0.05) ** 2).sum(axis=1)
        return 0

    v0 = C.c_void_p(L.c3control_init_value(ctl.h, FIBER_FN(start), None, aa, 0))
    results, taken, tried = {}, 0, 0
    names = ("C3SC_HOST_CROSS", "C3SC_NO_SPECULATE", "C3SC_ALWAYS_SPECULATE")
    L.valuef_interp_counter.restype = C.c_size_t
    for path in ("speculative", "sequential", "host"):
        for var in names:
            os.environ.pop(var, None)
        if path == "host":
            os.environ["C3SC_HOST_CROSS"] = "1"
        elif path == "sequential":
            os.environ["C3SC_NO_SPECULATE"] = "1"
        else:  # try it in EVERY warm-started sweep, not only after a sweep that kept its sets: the failing attempts are the hard case
            os.environ["C3SC_ALWAYS_SPECULATE"] = "1"
        before = (L.valuef_interp_counter(0), L.valuef_interp_counter(1))
        v = C.c_void_p(L.valuef_copy(v0))
        rows = []
        for upd in range(6):
            diag = C.c_void_p(None)
            nxt = C.c_void_p(L.c3control_pi_solve(ctl.h, C.c_size_t(8), C.c_double(1e-12), v, aa, ctl.opt, 0, C.byref(diag)))
            L.valuef_destroy(v)
            v = C.c_void_p(L.c3control_vi_solve(ctl.h, C.c_size_t(2), C.c_double(1e-12), nxt, aa, ctl.opt, 0, C.byref(diag)))
            L.valuef_destroy(nxt)
            rows.append((L.diag_count(diag), L.diag_last_diff(diag)) + _cores_of(L, v, w))
            L.diag_destroy(C.byref(diag))
        results[path] = rows
        L.valuef_destroy(v)
        if path == "speculative":
            tried, taken = L.valuef_interp_counter(0) - before[0], L.valuef_interp_counter(1) - before[1]
        else:
            assert L.valuef_interp_counter(0) == before[0]
    for var in names:
        os.environ.pop(var, None)
    for other in ("sequential", "host"):
        for it, (a, b) in enumerate(zip(results["speculative"], results[other])):
            assert a[0] == b[0] and a[1] == b[1], f"{other}, update {it}: sweeps / last step {a[:2]} vs {b[:2]}"
            assert a[2] == b[2], f"{other}, update {it}: ranks {a[2]} vs {b[2]}"
            for m in range(d):
                assert np.array_equal(a[3][m], b[3][m]), f"{other}, update {it}, core {m}: max diff {np.abs(a[3][m] - b[3][m]).max():.3e}"
    print(f"{name} {w.ngrid}: 6 control updates (60 sweeps) bit-identical with and without the speculative first iteration and on the "
          f"host driver; short cut tried in {tried} sweeps, confirmed in {taken}")
    assert taken >= 1 or not must_take
    L.valuef_destroy(v0)
    L.approx_args_free(aa)
    ctl.close()


@pytest.mark.parametrize("name,kw,aargs", [
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=2)),
    ("car7d", dict(), dict(maxrank=10, kick=2)),                       # the bench's vi_sweep configuration: 41^7, rank cap 10
    ("dubins3d", dict(ngrid=(41, 41, 41), rank=4), dict(maxrank=12, kick=3)),
    ("lqg2d", dict(ngrid=(60, 60), rank=4), dict(maxrank=20, kick=5)),  # core steps of up to 60 x 20 x 20: the factorisation leaves LDS
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=2, crossrank=10)),  # cross at rank 10, rounded to 5
    ("car7d", dict(), dict(maxrank=10, kick=4, crossrank=20)),          # cross rank 20 -> 10 at full size (the matrices still fit LDS)
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=9, crossrank=40)),  # ranks above 32: global-scratch core steps
    ("car7d", dict(), dict(maxrank=10, kick=10, crossrank=26)),         # 1066 x 26 at full size: the global-scratch core step (left-looking LU, rows in registers) on real matrices
    ("car7d", dict(), dict(maxrank=10, kick=40, crossrank=48, cross_maxiter=1)),  # vi_iters_to_tol's configuration: 1968 x 48 and 1681 x 41 (padded to 48) steps
    ("dubins3d", dict(ngrid=(101, 101, 101), rank=4), dict(maxrank=8, kick=2)),  # 808 x 8: two rows per thread in the register core step
    ("dubins3d", dict(ngrid=(101, 101, 101), rank=4), dict(maxrank=12, kick=12, crossrank=24)),  # 2424 x 24: more rows than the register panels hold (the plain left-looking LU of the global-scratch step)
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=3, crossrank=12, cross_maxiter=1)),  # one cross iteration per sweep
], ids=["car7d-small", "car7d-41", "dubins3d", "lqg2d", "car7d-small-crossrank10", "car7d-41-crossrank20", "car7d-small-crossrank40",
        "car7d-41-crossrank26", "car7d-41-crossrank48",
        "dubins3d-101-rank8", "dubins3d-101-crossrank24", "car7d-small-one-cross-iteration"])
def test_device_resident_cross_iterations_match_the_host_driver(name, kw, aargs):
    """c3control_step_vi with whole cross iterations on the device (c3sc_hip_cross_*: fiber index lists, Bellman launches, node
    memo, pivoted factorisation + maxvol of every core step on one stream) against the same sweeps driven from the host
    (C3SC_HOST_CROSS=1: index lists, memo and factorisations in c3sc_cross.c / c3sc_bellman.c, one kernel launch per core step).
    lu_maxvol and its device twin reorder no floating-point sum and search pivots with exact integer keys: cores, ranks and the
    number of node evaluations must be IDENTICAL, bit for bit, sweep after sweep."""
    w = wl.WORKLOADS[name]().scaled(**kw) if kw else wl.WORKLOADS[name]()
    L, fl, ctl, aa = _setup(w, **aargs)
    d = w.dx

    def start(n, x, out, a):
        X = np.ctypeslib.as_array(x, shape=(n, d))
        np.ctypeslib.as_array(out, shape=(n,))[:] = 1.0 + 0.1 * ((X - 0.05) ** 2).sum(axis=1)
        return 0

    v0 = C.c_void_p(L.c3control_init_value(ctl.h, FIBER_FN(start), None, aa, 0))
    results = {}
    for path in ("device", "host"):
        if path == "host":
            os.environ["C3SC_HOST_CROSS"] = "1"
        else:
            os.environ.pop("C3SC_HOST_CROSS", None)
        v = C.c_void_p(L.valuef_copy(v0))
        rows = []
        ne = C.c_size_t(0)
        for it in range(4):
            nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))
            L.valuef_destroy(v)
            v = nxt
            ranks, cores = _cores_of(L, v, w)
            rows.append((ne.value, ranks, cores))
        results[path] = rows
        L.valuef_destroy(v)
    os.environ.pop("C3SC_HOST_CROSS", None)
    for it, (a, b) in enumerate(zip(results["device"], results["host"])):
        assert a[1] == b[1], f"sweep {it}: ranks {a[1]} vs {b[1]}"
        assert a[0] == b[0], f"sweep {it}: node evaluations {a[0]} vs {b[0]}"
        for m in range(d):
            assert np.array_equal(a[2][m], b[2][m]), f"sweep {it}, core {m}: max diff {np.abs(a[2][m] - b[2][m]).max():.3e}"
    print(f"{name} {w.ngrid}: 4 sweeps, device-resident and host-driven cross iterations bit-identical; ranks {results['device'][-1][1]}, "
          f"node evaluations per sweep {[r[0] for r in results['device']]}")
    L.valuef_destroy(v0)
    L.approx_args_free(aa)
    ctl.close()


@pytest.mark.parametrize("name,kw,aargs", [
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=2)),
    ("dubins3d", dict(ngrid=(41, 41, 41), rank=4), dict(maxrank=12, kick=3)),
    ("lqg2d", dict(ngrid=(50, 50), rank=4), dict(maxrank=20, kick=5)),
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=3, crossrank=12)),                   # policy evaluation with the cross above the cap
    ("car7d", dict(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4), dict(maxrank=5, kick=3, crossrank=12, cross_maxiter=1)),  # ... and one cross iteration per sweep
], ids=["car7d-small", "dubins3d", "lqg2d", "car7d-small-crossrank12", "car7d-small-crossrank12-one-cross-iteration"])
def test_device_resident_policy_iteration_matches_the_host_driver(name, kw, aargs):
    """The examples' control update -- c3control_pi_solve (10 evaluation sweeps of one policy, bellman_pi) followed by one
    c3control_vi_solve step -- with whole cross iterations on the device (policy pass with the per-node policy memo, evaluation
    pass, factorisations) against the host-driven path: cores, ranks and both evaluation counters identical, bit for bit."""
    w = wl.WORKLOADS[name]().scaled(**kw)
    L, fl, ctl, aa = _setup(w, **aargs)
    d = w.dx

    def start(n, x, out, a):
        X = np.ctypeslib.as_array(x, shape=(n, d))
        np.ctypeslib.as_array(out, shape=(n,))[:] = 1.0 + 0.1 * ((X - 0.05) ** 2).sum(axis=1)
        return 0

    v0 = C.c_void_p(L.c3control_init_value(ctl.h, FIBER_FN(start), None, aa, 0))
    results = {}
    for path in ("device", "host"):
        if path == "host":
            os.environ["C3SC_HOST_CROSS"] = "1"
        else:
            os.environ.pop("C3SC_HOST_CROSS", None)
        v = C.c_void_p(L.valuef_copy(v0))
        rows = []
        for upd in range(3):
            diag = C.c_void_p(None)
            nxt = C.c_void_p(L.c3control_pi_solve(ctl.h, C.c_size_t(6), C.c_double(1e-9), v, aa, ctl.opt, 0, C.byref(diag)))
            L.valuef_destroy(v)
            v = C.c_void_p(L.c3control_vi_solve(ctl.h, C.c_size_t(1), C.c_double(1e-9), nxt, aa, ctl.opt, 0, C.byref(diag)))
            L.valuef_destroy(nxt)
            rows.append((L.diag_count(diag), L.diag_last_diff(diag)) + _cores_of(L, v, w))
            L.diag_destroy(C.byref(diag))
        results[path] = rows
        L.valuef_destroy(v)
    os.environ.pop("C3SC_HOST_CROSS", None)
    for it, (a, b) in enumerate(zip(results["device"], results["host"])):
        assert a[0] == b[0] and a[1] == b[1], f"update {it}: sweeps / last step {a[:2]} vs {b[:2]}"
        assert a[2] == b[2], f"update {it}: ranks {a[2]} vs {b[2]}"
        for m in range(d):
            assert np.array_equal(a[3][m], b[3][m]), f"update {it}, core {m}: max diff {np.abs(a[3][m] - b[3][m]).max():.3e}"
    print(f"{name} {w.ngrid}: 3 control updates (6 policy-evaluation sweeps + 1 value-iteration sweep each): device-resident and host-driven "
          f"paths bit-identical; ranks {results['device'][-1][2]}")
    L.valuef_destroy(v0)
    L.approx_args_free(aa)
    ctl.close()


def test_car7d_free_running_solves_device_vs_oracle(oracle):
    """Free-running (not lock-step): the SAME 12 control updates of the reduced 7-D car solved independently on the device path
    (device-resident cross iterations, one-launch confirmations, policy iteration on the device) and on the oracle-fed host path,
    each from its own previous state.  The cross drivers are bit-identical twins and the backups agree to ~1e-15, so the two
    trajectories stay together: nodal L-inf difference after every update within 1e-6 of max |V| (north_star's statement
    "value-function L-inf error within 1e-6 of reference after N iterations" on a car7d configuration)."""
    import regression_lib as R

    w0 = wl.c4_car7d().scaled(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4)
    cands = np.array([[a, b] for a in (-0.5, 0.07, 0.43) for b in (-1.0, 0.13, 0.91)])
    w = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, w0.ranks, w0.discount, w0.bc, list(w0.obstacles), cands)
    wts = np.array([0.3, 0.5, 0.2, 0.1, 0.15, 0.7, 0.25])
    cfg = dict(w=w, max_updates=12, conv=1e-9, adapt=1, startrank=3, maxrank=5, kick=2, cross_tol=1e-10, round_tol=1e-9,
               start_fn=lambda X: 1.0 + ((X - 0.1) ** 2 * wts).sum(axis=1))
    gpu, orc = R.GpuLoop(cfg), R.OracleLoop(cfg)
    diffs = []
    a, b = gpu.init_value(), orc.init_value()
    for _ in range(12):
        a = gpu.run(max_updates=1, cost=a)
        b = orc.run(max_updates=1, cost=b)
        vb = orc.nodal(b)
        diffs.append(np.abs(gpu.nodal(a) - vb).max() / np.abs(vb).max())
    print("free-running, per update:", " ".join(f"{x:.1e}" for x in diffs))
    assert max(diffs) <= 1e-6
    gpu.L.valuef_destroy(a)
    gpu.L.valuef_destroy(b)
    gpu.close()
    orc.close()


_OVERFLOW_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import facade_lib
from c3sc_amd import workloads as wl
L = facade_lib.lib()
for n in ("c3control_init_value", "c3control_step_vi"):
    getattr(L, n).restype = C.c_void_p
L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
w = wl.c4_car7d().scaled(ngrid=(13, 12, 14, 11, 10, 9, 15), rank=4)
ctl = facade_lib.Control(w, consistent_ends={ce})
aa = C.c_void_p(L.approx_args_init())
L.approx_args_set_maxrank(aa, C.c_size_t(8)); L.approx_args_set_startrank(aa, C.c_size_t(3)); L.approx_args_set_kickrank(aa, C.c_size_t(2))
L.approx_args_set_cross_tol(aa, C.c_double(1e-8)); L.approx_args_set_round_tol(aa, C.c_double(1e-8))
d = w.dx
start = facade_lib.FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).__setitem__(slice(None), 1.0 + 0.1 * (np.ctypeslib.as_array(x, shape=(n, d)) ** 2).sum(axis=1)), 0)[1])
v = C.c_void_p(L.c3control_init_value(ctl.h, start, None, aa, 0))
ne = C.c_size_t(0)
for _ in range(4):
    nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))
    L.valuef_destroy(v); v = nxt
ranks = [int(L.valuef_get_ranks(v)[i]) for i in range(d + 1)]
L.valuef_norm.restype = C.c_double
print("RESULT", ranks, repr(L.valuef_norm(v)))
"""


@pytest.mark.gpu
def test_device_memo_overflow_grows_the_table_and_gives_the_same_cores():
    """The device node memo filling up used to end the solve.  Under consistent end points a fiber value is a function of
    its node (up to the last bits: the train is contracted in another order along another direction), so values computed while the
    table was full are the ones a larger table would have returned: the table is doubled (c3sc_hip_cross_grow_memo) and the sweep
    goes on -- same ranks, |V| within the rank-8 approximation's own spread of the run with an ample table (last-bit differences in
    fiber values may flip pivots at this rank cap).  Under the reference's literal end-point rule the memo is part of
    the semantics (first stored value wins): the solve still stops."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(ce, small):
        env = dict(os.environ)
        if small:
            env.update(C3SC_MEMO_MIN_LOG2="10", C3SC_MEMO_SCALE="0")
        return subprocess.run([sys.executable, "-c", _OVERFLOW_SCRIPT.format(root=root, ce=ce)], env=env, capture_output=True, text=True, timeout=600)

    ample, tiny = run("True", False), run("True", True)
    assert ample.returncode == 0 and tiny.returncode == 0, (tiny.returncode, tiny.stderr[-1500:])
    assert "memo was full" in tiny.stderr and "memo was full" not in ample.stderr
    ra = [ln for ln in ample.stdout.splitlines() if ln.startswith("RESULT")][0]
    rt = [ln for ln in tiny.stdout.splitlines() if ln.startswith("RESULT")][0]
    assert ra.split("]")[0] == rt.split("]")[0], (ra, rt)  # ranks
    na, nt = float(ra.split("]")[1]), float(rt.split("]")[1])
    print(f"|V| with an ample memo {na:.6f}, after overflow + growth {nt:.6f}")
    assert abs(na - nt) <= 2e-2 * na
    lit = run("False", True)
    assert lit.returncode == 1 and "memo overflowed" in lit.stderr


@pytest.mark.gpu
def test_streamed_orthogonalisation_returns_the_same_bits_as_the_plain_hand_over():
    """The last cross iteration of an interpolation streams its cores to the host as the right-to-left steps finish them and the
    rounding's orthogonalisation factors each while the device works on the next (c3sc_hip_cross_iteration_streamed,
    c3sc_hip_cross_wait_core); C3SC_NO_STREAMED_ROUNDING=1 fetches the finished train and rounds it afterwards.  Same arithmetic:
    cores, ranks and node counts of six sweeps at an elevated cross rank must be identical, bit for bit."""
    w = wl.WORKLOADS["car7d"]()
    results = {}
    for mode in ("streamed", "plain"):
        if mode == "plain":
            os.environ["C3SC_NO_STREAMED_ROUNDING"] = "1"
        else:
            os.environ.pop("C3SC_NO_STREAMED_ROUNDING", None)
        try:
            L, fl, ctl, aa = _setup(w, maxrank=10, kick=10, crossrank=30, cross_maxiter=1)
            d = w.dx

            def start(n, x, out, a):
                X = np.ctypeslib.as_array(x, shape=(n, d))
                np.ctypeslib.as_array(out, shape=(n,))[:] = 1.0 + 0.1 * ((X - 0.05) ** 2).sum(axis=1)
                return 0

            v = C.c_void_p(L.c3control_init_value(ctl.h, FIBER_FN(start), None, aa, 0))
            rows = []
            ne = C.c_size_t(0)
            for it in range(6):
                nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))
                L.valuef_destroy(v)
                v = nxt
                ranks, cores = _cores_of(L, v, w)
                rows.append((ne.value, ranks, cores))
            results[mode] = rows
            L.valuef_destroy(v)
            L.approx_args_free(aa)
            ctl.close()
        finally:
            os.environ.pop("C3SC_NO_STREAMED_ROUNDING", None)
    for a, b in zip(results["streamed"], results["plain"]):
        assert a[0] == b[0] and a[1] == b[1], (a[0], b[0], a[1], b[1])
        for ca, cb in zip(a[2], b[2]):
            assert ca.tobytes() == cb.tobytes()
    print(f"car7d 41^7, cross rank 30 -> 10: six sweeps streamed and plain bit-identical; node evaluations {[r[0] for r in results['streamed']]}")
