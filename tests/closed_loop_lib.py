"""Shared pieces of the closed-loop replays of the reference's Test_bellman_vi (tprob_test.c:1817-1897) and
Test_bellman_pi3d (:2448-2540): problem set-up, the value-iteration loop with a log, and the closed loop run_sim_2d_1d /
run_sim_3d_3d (:61-113) restated -- cdyn's rk4 integrator with the controller inside the right-hand side, integrator step
1e-3 (2-D) / 1e-2 (3-D), as plain numpy."""
import ctypes as C

import numpy as np

import regression_lib as R
from c3sc_amd import workloads as wl

VI_CANDS = np.linspace(-3.0, 3.0, 49).reshape(-1, 1)  # the oracle path's stand-in for BFGS on [-3, 3] (tprob_test.c:1832-1842)


def vi_workload():
    w = R.workload(100)  # tprob_test.c:1824-1830: [-2,2]^2, 100 x 100, discount 0.1
    return wl.Workload(w.name, w.model, w.params, w.dx, w.du, w.lb, w.ub, w.ngrid, w.ranks, w.discount, w.bc, [], VI_CANDS)


def vi_cfg():
    # tprob_test.c:1845-1851; maxrank 30 there -- 20 is the largest rank the device kernels serve (c3sc_hip_max_rank), used on both paths
    return dict(w=vi_workload(), max_updates=1, conv=1e-5, adapt=1, startrank=2, maxrank=20, kick=5, cross_tol=1e-8, round_tol=1e-8,
                start_value=0.2, box=([-3.0], [3.0]))


def vi_loop(path, minimiser="bruteforce", callbacks=None):
    return R.OracleLoop(vi_cfg()) if path == "oracle" else R.GpuLoop(vi_cfg(), minimiser, callbacks)


def vi_solve_logged(loop, max_sweeps, tol, every=0):
    """c3control_vi_solve (bellman.c:2282-2340) one sweep at a time so that the step sizes can be logged; the memo is reset as
    the reference does (every 1000 iterations) because each call of vi_solve(1) resets it anyway."""
    L = loop.L
    cur = loop.init_value()
    hist = []
    for ii in range(max_sweeps):
        nxt = loop.vi_solve(1, tol, cur)
        diff = L.valuef_norm2diff(cur, nxt)
        L.valuef_destroy(cur)
        cur = nxt
        hist.append((ii, diff, L.valuef_norm(cur), loop.rank(cur)))
        if every and ii % every == 0:
            print(f"sweep {ii:6d}  step {diff:.6e}  |V| {hist[-1][2]:.9f}  rank {hist[-1][3]}", flush=True)
        if diff < tol:
            break
    return cur, hist


def simulate_rk4(drift, controller, x0, t_final, dt_outer, dt_int):
    """trajectory_step(traj, ode_sys, dt_outer) until t_final with cdyn's "rk4" at step dt_int; the controlled right-hand side
    calls the controller at every stage (integrator_create_controlled)."""
    x = np.array(x0, dtype=np.float64)
    t = 0.0

    def rhs(tt, xx):
        return drift(xx, controller(tt, xx))

    nsub = int(round(dt_outer / dt_int))
    while t < t_final:
        for _ in range(nsub):
            k1 = rhs(t, x)
            k2 = rhs(t + dt_int / 2, x + dt_int / 2 * k1)
            k3 = rhs(t + dt_int / 2, x + dt_int / 2 * k2)
            k4 = rhs(t + dt_int, x + dt_int * k3)
            x = x + dt_int / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
            t += dt_int
    return x


def oracle_controller(oracle, P, cands):
    lib = oracle.lib()

    def ctl(t, x):
        xx = np.ascontiguousarray(np.clip(x, P_lb(P), P_ub(P)))
        ui, val = C.c_int(-5), C.c_double(0.0)
        rc = lib.orc_policy_eval(P.h, oracle.dp(xx), C.byref(ui), C.byref(val))
        assert rc == 0
        return cands[ui.value] if ui.value >= 0 else np.zeros(cands.shape[1])

    return ctl


def P_lb(P):
    return np.array(P.w.lb)


def P_ub(P):
    return np.array(P.w.ub)


# ---- Test_bellman_pi3d (tprob_test.c:2448-2540)
def pi3d_cfg():
    w = wl.WORKLOADS["tprob3d"]()  # 25^3, rank 10, beta 0.1, absorbing faces; candidates: 5^3 over [-5,5]^3
    return dict(w=w, max_updates=400, conv=1e-3, adapt=0, startrank=10, maxrank=10, kick=10, cross_tol=1e-8, round_tol=1e-7, pi_sweeps=20,
                break_on_conv=True, start_fn=lambda X: (X ** 2).sum(axis=1), box=([-5.0] * 3, [5.0] * 3))  # quad3d, :1567-1574


def pi3d_loop(path, minimiser="bruteforce", callbacks=None):
    return R.OracleLoop(pi3d_cfg()) if path == "oracle" else R.GpuLoop(pi3d_cfg(), minimiser, callbacks)


def f3(x, u):  # tprob_test.c:223-251
    return np.array([x[0] * x[2] ** 2 * u[0], -x[1] * u[2] + u[1], x[0] * x[1] * u[0] + 2 * u[1]])
