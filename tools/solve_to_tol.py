"""The reference examples' outer loop (e.g. dubinscar.c:343-352): pi_solve(10, 1e-2) then one vi_solve step, repeated
until the value-iteration step moves the value function by less than abs_conv_vi in L2 -- "VI iterations to tolerance"
with wall time, through libc3sc.so on the GPU.
    python tools/solve_to_tol.py [dubins3d] [ngrid] [maxrank] [tol] [max_outer]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import facade_lib  # noqa: E402
from c3sc_amd import workloads as wl  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dubins3d"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 101
maxrank = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
max_outer = int(sys.argv[5]) if len(sys.argv) > 5 else 300
w = wl.WORKLOADS[name]().scaled(ngrid=(n,) * wl.WORKLOADS[name]().dx)
L = facade_lib.lib()
for f in ("c3control_init_value", "c3control_vi_solve", "c3control_pi_solve"):
    getattr(L, f).restype = C.c_void_p
for f in ("valuef_norm", "valuef_norm2diff"):
    getattr(L, f).restype = C.c_double
L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
L.diag_count.restype = C.c_size_t
ctl = facade_lib.Control(w, consistent_ends=None)  # the library default: consistent end points
aa = C.c_void_p(L.approx_args_init())
L.approx_args_set_cross_tol(aa, C.c_double(1e-5))
L.approx_args_set_round_tol(aa, C.c_double(1e-5))
L.approx_args_set_kickrank(aa, C.c_size_t(5))
L.approx_args_set_startrank(aa, C.c_size_t(5))
L.approx_args_set_maxrank(aa, C.c_size_t(maxrank))
d = w.dx
start = facade_lib.FIBER_FN(lambda N, x, out, a: (np.ctypeslib.as_array(out, shape=(N,)).fill(0.0), 0)[1])
cost = C.c_void_p(L.c3control_init_value(ctl.h, start, None, aa, 0))
diag = C.c_void_p(None)
t0 = time.perf_counter()
for ii in range(max_outer):
    nxt = C.c_void_p(L.c3control_pi_solve(ctl.h, C.c_size_t(10), C.c_double(1e-2), cost, aa, ctl.opt, 0, C.byref(diag)))
    L.valuef_destroy(cost)
    cost = C.c_void_p(L.c3control_vi_solve(ctl.h, C.c_size_t(1), C.c_double(tol), nxt, aa, ctl.opt, 0, C.byref(diag)))
    diff = L.valuef_norm2diff(nxt, cost)
    L.valuef_destroy(nxt)
    ranks = [L.valuef_get_ranks(cost)[i] for i in range(d + 1)]
    if ii < 5 or ii % 10 == 0 or diff < tol:
        print(f"outer {ii:4d}: |V_vi - V_pi|_L2 = {diff:.4e}  |V| = {L.valuef_norm(cost):.4e}  ranks {ranks}  "
              f"t = {time.perf_counter() - t0:.2f} s  sweeps so far {L.diag_count(diag)}", flush=True)
    if diff < tol:
        print(f"CONVERGED: {ii + 1} outer iterations, {L.diag_count(diag)} Bellman sweeps, {time.perf_counter() - t0:.2f} s")
        break
else:
    print(f"not converged after {max_outer} outer iterations ({time.perf_counter() - t0:.1f} s), last diff {diff:.3e}")
