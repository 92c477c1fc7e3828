"""Whole value-iteration sweeps through the reference API (c3control_step_vi -> own cross driver -> batched kernels):
wall time per sweep, fibers and node backups requested, and the CPU oracle's time for the same node count.
    python tools/vi_step_bench.py [car7d|dubins3d|lqg6d] [sweeps]
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

name = sys.argv[1] if len(sys.argv) > 1 else "car7d"
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = wl.WORKLOADS[name]()
L = facade_lib.lib()
for n in ("c3control_init_value", "c3control_step_vi"):
    getattr(L, n).restype = C.c_void_p
for n in ("valuef_norm", "valuef_norm2diff"):
    getattr(L, n).restype = C.c_double
L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
ctl = facade_lib.Control(w)
aa = C.c_void_p(L.approx_args_init())
L.approx_args_set_cross_tol(aa, C.c_double(1e-6))
L.approx_args_set_round_tol(aa, C.c_double(1e-5))
L.approx_args_set_kickrank(aa, C.c_size_t(2))
L.approx_args_set_startrank(aa, C.c_size_t(4))
L.approx_args_set_maxrank(aa, C.c_size_t(max(w.ranks)))
FIBER_FN = C.CFUNCTYPE(C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
d = w.dx


def start(n, x, out, a):
    X = np.ctypeslib.as_array(x, shape=(n, d))
    np.ctypeslib.as_array(out, shape=(n,))[:] = 1.0 + 0.1 * (X ** 2).sum(axis=1)
    return 0


cb = FIBER_FN(start)
vf = C.c_void_p(L.c3control_init_value(ctl.h, cb, None, aa, 0))
ne = C.c_size_t(0)
rows = []
for it in range(sweeps):
    t0 = time.perf_counter()
    nxt = C.c_void_p(L.c3control_step_vi(ctl.h, vf, aa, ctl.opt, 0, C.byref(ne)))
    dt = time.perf_counter() - t0
    diff, norm = L.valuef_norm2diff(vf, nxt), L.valuef_norm(nxt)
    ranks = [L.valuef_get_ranks(nxt)[i] for i in range(d + 1)]
    rows.append((it, dt, ne.value, diff, norm, ranks))
    print(f"sweep {it}: {dt*1e3:8.1f} ms  node backups {ne.value:8d}  ({ne.value/dt:.3e} nodes/s through the driver)  "
          f"|dV| {diff:.3e} |V| {norm:.3e} ranks {ranks}", flush=True)
    L.valuef_destroy(vf)
    vf = nxt
print("mean ms per sweep:", 1e3 * np.mean([r[1] for r in rows[1:] or rows]))
