"""Pure value iteration to tolerance through libc3sc.so: c3control_vi_solve's loop (bellman.c:2282-2340: stop when the L2 step
||V_{i+1} - V_i|| falls below abs_conv_tol), one sweep per call so that the step series can be logged.
    python tools/vi_to_tol.py [workload] [maxrank] [abs_tol] [max_sweeps] [ngrid]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "car7d"
maxrank = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
max_sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
w = wl.WORKLOADS[name]()
if len(sys.argv) > 5:
    w = w.scaled(ngrid=(int(sys.argv[5]),) * w.dx)
L = facade_lib.lib()
for f in ("c3control_init_value", "c3control_step_vi"):
    getattr(L, f).restype = C.c_void_p
for f in ("valuef_norm", "valuef_norm2diff"):
    getattr(L, f).restype = C.c_double
L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
ctl = facade_lib.Control(w, consistent_ends=None)
aa = C.c_void_p(L.approx_args_init())
L.approx_args_set_cross_tol(aa, C.c_double(1e-6))
L.approx_args_set_round_tol(aa, C.c_double(1e-6))
L.approx_args_set_kickrank(aa, C.c_size_t(2))
L.approx_args_set_startrank(aa, C.c_size_t(4))
L.approx_args_set_maxrank(aa, C.c_size_t(maxrank))
d = w.dx
zero = facade_lib.FIBER_FN(lambda N, x, out, a: (np.ctypeslib.as_array(out, shape=(N,)).fill(0.0), 0)[1])
v = C.c_void_p(L.c3control_init_value(ctl.h, zero, None, aa, 0))
ne = C.c_size_t(0)
t0 = time.perf_counter()
nodes = 0
for ii in range(max_sweeps):
    nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))
    nodes += ne.value
    diff = L.valuef_norm2diff(v, nxt)
    L.valuef_destroy(v)
    v = nxt
    if ii < 5 or ii % max(1, max_sweeps // 50) == 0 or diff < tol:
        print(f"sweep {ii:6d}: step {diff:.4e}  |V| {L.valuef_norm(v):.6e}  ranks {[int(L.valuef_get_ranks(v)[i]) for i in range(d + 1)]}  "
              f"t = {time.perf_counter() - t0:.2f} s  node backups so far {nodes}", flush=True)
    if diff < tol:
        print(f"CONVERGED: {ii + 1} value-iteration sweeps to |V_i+1 - V_i|_L2 < {tol:g} in {time.perf_counter() - t0:.2f} s")
        break
else:
    print(f"not converged after {max_sweeps} sweeps ({time.perf_counter() - t0:.1f} s), last step {diff:.3e}")
