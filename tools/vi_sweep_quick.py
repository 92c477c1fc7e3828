"""Times c3control_step_vi on a workload through libc3sc.so (device-resident or host-driven cross iterations):
    python tools/vi_sweep_quick.py [workload] [sweeps]        C3SC_HOST_CROSS=1 for the host-driven path, C3SC_PROFILE=1 for the breakdown"""
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
from c3sc_amd.engine import load_library  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "car7d"
nsweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = load_library()
L = facade_lib.lib()
for n in ("c3control_init_value", "c3control_step_vi"):
    getattr(L, n).restype = C.c_void_p
L.valuef_norm.restype = C.c_double
L.valuef_norm2diff.restype = C.c_double
L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
w = wl.WORKLOADS[name]()
d = w.dx
ctl = facade_lib.Control(w, consistent_ends=None)
aa = C.c_void_p(L.approx_args_init())
L.approx_args_set_cross_tol(aa, C.c_double(1e-6))
L.approx_args_set_round_tol(aa, C.c_double(1e-5))
L.approx_args_set_kickrank(aa, C.c_size_t(2))
L.approx_args_set_startrank(aa, C.c_size_t(4))
L.approx_args_set_maxrank(aa, C.c_size_t(max(w.ranks)))


def smooth(n, x, out, a):
    X = np.ctypeslib.as_array(x, shape=(n, d))
    np.ctypeslib.as_array(out, shape=(n,))[:] = 1.0 + 0.1 * (X ** 2).sum(axis=1)
    return 0


vf = C.c_void_p(L.c3control_init_value(ctl.h, facade_lib.FIBER_FN(smooth), None, aa, 0))
ne = C.c_size_t(0)
for it in range(nsweeps):
    l0, t0 = H.c3sc_hip_launch_count(), time.perf_counter()
    nxt = C.c_void_p(L.c3control_step_vi(ctl.h, vf, aa, ctl.opt, 0, C.byref(ne)))
    dt = time.perf_counter() - t0
    print(f"sweep {it}: {1e3 * dt:7.3f} ms  {ne.value:7d} node backups  {ne.value / dt:.3e} nodes/s  launches {H.c3sc_hip_launch_count() - l0:3d}  "
          f"step {L.valuef_norm2diff(vf, nxt):.4e}  |V| {L.valuef_norm(nxt):.6e}  ranks {[int(L.valuef_get_ranks(nxt)[i]) for i in range(d + 1)]}", flush=True)
    L.valuef_destroy(vf)
    vf = nxt
