"""off-grid evaluation, the implicit policy (c3control_policy_eval / controller), the closed-loop
Euler tail and value-function files -- host code, checked against the oracle's restatement of nodeutil.c:718-816 and
bellman.c:2105-2158.  No GPU involved (a single state is not GPU work; the reference does the same on the host)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from c3sc_amd import workloads as wl  # noqa: E402
from test_facade import _callbacks  # noqa: E402


def _setup():
    import facade_lib

    L = facade_lib.lib()
    L.valuef_eval.restype = C.c_double
    L.valuef_norm2diff.restype = C.c_double
    L.valuef_load.restype = C.c_void_p
    L.valuef_loadtxt.restype = C.c_void_p
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_eval_ind.restype = C.c_double
    return L, facade_lib


def test_offgrid_stencil_and_policy_eval_match_oracle(oracle):
    L, fl = _setup()
    w = wl.c2_dubins().scaled(ngrid=(21, 17, 16), rank=4)
    cores = wl.synth_cores(w)
    P = oracle.Problem(w, cores)
    ctl = fl.Control(w, _callbacks(w), device_model=False)
    vf = ctl.valuef(cores)
    xg = ctl.xgrid()
    gs = [fl.f64(g) for g in xg]
    L.valuef_attach_grid(vf, fl.ptrs(gs))
    L.c3control_add_policy_sim(ctl.h, vf, ctl.opt, None)
    lib = oracle.lib()
    lib.orc_valuef_eval.restype = C.c_double
    gp = oracle.ptr_array(gs)
    rng = np.random.default_rng(5)
    lo, hi = np.array(w.lb), np.array(w.ub)
    pts = [lo + (hi - lo) * rng.uniform(0, 1, w.dx) for _ in range(60)]
    pts += [np.array([lo[0] + 1e-3, 0.3, hi[2] - 1e-3]), np.array([hi[0] - 1e-3, lo[1] + 1e-3, lo[2] + 1e-3]),
            np.array([0.1, -0.1, 0.2])]  # near faces (absorb / periodic wrap) and inside the obstacle
    S = 2 * w.dx + 1
    for x in pts:
        x = fl.f64(x)
        got, ab = np.full(S, -7.0), C.c_int(9)
        want, ab0 = np.full(S, -7.0), C.c_int(9)
        rc = L.mca_get_neighbor_node_costs(C.c_size_t(w.dx), fl.dp(x), ctl.bound(), vf, fl.sp(fl.usz(w.ngrid)), fl.ptrs(gs),
                                           C.byref(ab), fl.dp(got))
        rc0 = lib.orc_mca_get_neighbor_node_costs(C.c_size_t(w.dx), oracle.dp(x), P.boundary_handle(), P.vf.h,
                                                  oracle.sp(oracle.usz(w.ngrid)), gp, C.byref(ab0), oracle.dp(want))
        assert rc == 0 and rc0 == 0 and ab.value == ab0.value
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)
        assert L.valuef_eval(vf, fl.dp(x)) == pytest.approx(lib.orc_valuef_eval(P.vf.h, gp, oracle.dp(x)), rel=1e-13)
        # greedy control at the state: same candidate as the oracle's bellman_optimal restatement
        u = np.zeros(w.du)
        assert L.c3control_policy_eval(ctl.h, C.c_double(0.0), fl.dp(x), fl.dp(u)) == 0
        ui, val = C.c_int(-5), C.c_double(0.0)
        assert lib.orc_policy_eval(P.h, oracle.dp(x), C.byref(ui), C.byref(val)) == 0
        if ui.value >= 0:
            np.testing.assert_array_equal(u, w.cands[ui.value])
    # closed loop: deterministic Euler steps stay finite and start at x0
    x0 = fl.f64([1.5, -1.0, 0.3])
    traj, utraj = np.zeros((41, w.dx)), np.zeros((40, w.du))
    assert L.c3control_simulate(ctl.h, fl.dp(x0), C.c_double(0.05), C.c_size_t(40), None, fl.dp(traj), fl.dp(utraj)) == 0
    np.testing.assert_array_equal(traj[0], x0)
    assert np.isfinite(traj).all() and set(np.unique(utraj)) <= set(w.cands.ravel())
    # dubins: |velocity| = 1 -> every step moves dt in the plane
    np.testing.assert_allclose(np.hypot(*(traj[1:, :2] - traj[:-1, :2]).T), 0.05, rtol=1e-12)
    ctl.close()


def test_value_function_files_roundtrip(tmp_path):
    L, fl = _setup()
    w = wl.c2_dubins().scaled(ngrid=(9, 8, 7), rank=3)
    cores = wl.synth_cores(w)
    ctl = fl.Control(w)
    vf = ctl.valuef(cores)
    gs = [fl.f64(g) for g in ctl.xgrid()]
    L.valuef_attach_grid(vf, fl.ptrs(gs))
    Ng = fl.usz(w.ngrid)
    for save, load, name in ((L.valuef_save, L.valuef_load, "v.c3sc"), (L.valuef_savetxt, L.valuef_loadtxt, "v.txt")):
        path = str(tmp_path / name).encode()
        assert save(vf, path) == 0  # 0 = success, as the reference (valuefunc.c:231-236)
        back = C.c_void_p(load(path, fl.sp(Ng), fl.ptrs(gs)))
        assert back.value is not None
        assert [L.valuef_get_ranks(back)[i] for i in range(4)] == list(w.ranks)
        assert L.valuef_norm2diff(vf, back) <= 1e-13  # same cores (the norm of an exact zero difference is rounding noise)
        ind = fl.usz([3, 2, 5])
        assert L.valuef_eval_ind(vf, fl.sp(ind)) == L.valuef_eval_ind(back, fl.sp(ind))  # bit-identical values
        L.valuef_destroy(back)
    assert load(str(tmp_path / "missing").encode(), fl.sp(Ng), fl.ptrs(gs)) is None  # examples probe for a saved cost this way
    # loading onto a finer grid resamples the cores (function_train_create_nodal, valuefunc.c:252)
    fine = [np.linspace(g[0], g[-1], 2 * len(g) - 1) for g in gs]
    Nf = fl.usz([len(g) for g in fine])
    path = str(tmp_path / "v.c3sc").encode()
    vfine = C.c_void_p(L.valuef_load(path, fl.sp(Nf), fl.ptrs([fl.f64(g) for g in fine])))
    x = fl.f64([0.3, -0.2, 0.1])
    assert L.valuef_eval(vfine, fl.dp(x)) == pytest.approx(L.valuef_eval(vf, fl.dp(x)), rel=1e-12)
    L.valuef_destroy(vfine)
    ctl.close()


def _lqg2d_callbacks():
    """The regression's host callbacks (tprob_test.c:132-168 f1b / s1, :253-271 stagecost2d, :302-318 boundcost / ocost)."""
    import facade_lib

    def drift(t, x, u, out, jac, args):
        out[0], out[1] = x[1], u[0]
        return 0

    def diff(t, x, u, out, grad, args):
        out[0], out[1], out[2], out[3] = 1.0, 0.0, 0.0, 1.0
        return 0

    def stage(t, x, u, out, grad):
        out[0] = x[0] * x[0] + x[1] * x[1] + u[0] * u[0]
        return 0

    def bcost(t, x, out):
        out[0] = 100.0
        return 0

    def ocost(x, out):
        out[0] = 0.0
        return 0

    return (facade_lib.DYN_FN(drift), facade_lib.DYN_FN(diff), facade_lib.STAGE_FN(stage), facade_lib.BOUND_FN(bcost),
            facade_lib.OBS_FN(ocost))


@pytest.mark.gpu
def test_policy_tail_on_a_value_function_solved_on_the_device(oracle, tmp_path):
    """SURVEY.md 8f-4 end to end on the GPU box: a value function solved by the device loop (the reference's regression
    problem, tprob_test.c:1996-2119, cut short) goes through valuef_save / valuef_load, the implicit policy
    (c3control_add_policy_sim + c3control_policy_eval, bellman.c:2034-2158) is compared with the oracle's restatement at
    random states, and the closed loop (c3control_simulate) drives the state towards the origin."""
    import regression_lib as R

    L, fl = _setup()
    w = R.workload(25)
    gpu = R.GpuLoop("pi_25_const", callbacks=_lqg2d_callbacks())  # the tail evaluates the user's callbacks on the host, as the reference does
    cost = gpu.run(max_updates=150)
    # files
    path = str(tmp_path / "v.c3sc").encode()
    assert L.valuef_save(cost, path) == 0
    gs = [fl.f64(g) for g in gpu.ctl.xgrid()]
    back = C.c_void_p(L.valuef_load(path, fl.sp(fl.usz(w.ngrid)), fl.ptrs(gs)))
    assert back.value is not None and L.valuef_norm2diff(cost, back) <= 1e-12 * gpu.norm(cost)
    # implicit policy vs the oracle on the same cores
    ranks, cores = gpu.cores_of(cost)
    wr = wl.Workload(w.name, w.model, w.params, w.dx, w.du, w.lb, w.ub, w.ngrid, tuple(ranks), w.discount, w.bc, list(w.obstacles), w.cands)
    P = oracle.Problem(wr, [c.reshape(w.ngrid[m], -1) for m, c in enumerate(cores)])
    lib = oracle.lib()
    L.c3control_add_policy_sim(gpu.ctl.h, back, gpu.ctl.opt, None)
    rng = np.random.default_rng(11)
    same = 0
    for _ in range(80):
        x = fl.f64(rng.uniform(-1.9, 1.9, 2))
        u = np.zeros(1)
        assert L.c3control_policy_eval(gpu.ctl.h, C.c_double(0.0), fl.dp(x), fl.dp(u)) == 0
        ui, val = C.c_int(-5), C.c_double(0.0)
        assert lib.orc_policy_eval(P.h, oracle.dp(x), C.byref(ui), C.byref(val)) == 0
        same += int(ui.value >= 0 and u[0] == w.cands[ui.value, 0])
    assert same >= 78  # the odd state may sit on a tie between neighbouring candidates
    # closed loop from (1.5, -1): the controller brings the state closer to the origin
    NS = 300
    x0 = fl.f64([1.5, -1.0])
    traj, us = np.zeros((NS + 1, 2)), np.zeros((NS, 1))
    assert L.c3control_simulate(gpu.ctl.h, fl.dp(x0), C.c_double(0.01), C.c_size_t(NS), None, fl.dp(traj), fl.dp(us)) == 0
    assert np.isfinite(traj).all() and np.abs(us).max() <= 1.0
    assert np.hypot(*traj[-1]) < 0.6 * np.hypot(*x0)
    L.valuef_destroy(back)
    L.valuef_destroy(cost)
    gpu.close()
