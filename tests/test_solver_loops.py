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


def _setup(w, maxrank=12, startrank=3, round_tol=1e-9):
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
    L.approx_args_set_kickrank(aa, C.c_size_t(3))
    L.approx_args_set_startrank(aa, C.c_size_t(startrank))
    L.approx_args_set_maxrank(aa, C.c_size_t(maxrank))
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
    is compared node by node.  Most updates agree to ~1e-14 of max |V|; a few do not agree at all (0.1 ... 0.3): at a rank
    cap of 5 this value function is far from representable, the cross approximation's error is of that size, and which of
    several equally bad approximations comes out depends on pivot decisions that flip with the last bit of the fiber values
    -- the ORACLE-fed loop shows the same jumps against itself when 1e-16 relative noise is added to its fiber values
    (update 0 of this very configuration: 0.12).  So the statement tested is: the median update agrees to 1e-12 and at least
    three quarters of the updates to 1e-6; the outliers are printed.  The data avoid EXACT ties between candidates (the
    tie-break of the brute-force scan lives in C3): with the symmetric 3 x 3 candidate grid and a start value that does not
    depend on the steering / acceleration states, +u and -u tie at every node, the policy's pick among them is decided by
    the last bit (oracle: division per candidate; device: cross-multiplied comparison), and ten evaluation sweeps of two such
    policies drift apart by 20 % -- so the candidate list is slightly asymmetric and the start value depends on every
    coordinate."""
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
    good = [x for x in diffs if x <= 1e-6]
    print(f"car7d {w.ngrid}: 20 control updates in lock-step: {len(good)} agree to 1e-6 (worst of them {max(good):.1e}), median {np.median(diffs):.1e}, "
          f"pivot-flip outliers {[f'{x:.2f}' for x in diffs if x > 1e-6]}; |V| = {gpu.norm(state):.6e}, rank {gpu.rank(state)}")
    assert np.median(diffs) <= 1e-12 and len(good) >= 15 and max(good) <= 1e-12
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
