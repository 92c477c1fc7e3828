"""Create / use / destroy device contexts in a loop and watch free device memory and host RSS.
    python tools/leak_check.py [iterations]"""
import os, sys, resource
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = wl.c4_car7d().scaled(ngrid=(11,) * 7, rank=10)
cores = wl.synth_cores(w)
def free_mb(): return torch.cuda.mem_get_info()[0] / 2**20
def rss_mb(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
f0 = r0 = None
for it in range(n):
    eng = BellmanEngine(0); eng.configure(w, cores)
    for k in (0, 3, 6):
        idx = wl.synth_fibers(w, k, 50 if it % 2 else 40000)   # zero-copy and staged paths
        eng.bellman_fibers_host(k, idx)
    eng.close() if hasattr(eng, "close") else None
    del eng
    if it == 20: f0, r0 = free_mb(), rss_mb()
    if it % 50 == 0: print(it, "free MB", round(free_mb(), 1), "rss MB", round(rss_mb(), 1), flush=True)
print("after warm-up (iteration 20) -> end: device free", round(f0, 1), "->", round(free_mb(), 1), "MB; host max RSS", round(r0, 1), "->", round(rss_mb(), 1), "MB")

# ---- the host library: whole solver steps (controls, workspaces, value functions created and destroyed)
import ctypes as C
sys.path.insert(0, os.path.join(ROOT, "tests"))
import facade_lib as fl
L = fl.lib()
for f in ("c3control_init_value", "c3control_step_vi", "c3control_pi_solve", "c3control_vi_solve"):
    getattr(L, f).restype = C.c_void_p
w2 = wl.c1_lqg2d().scaled(ngrid=(21, 19))
start = fl.FIBER_FN(lambda N, x, out, a: (np.ctypeslib.as_array(out, shape=(N,)).fill(0.3), 0)[1])
m = max(20, n // 5)
for it in range(m):
    ctl = fl.Control(w2)
    aa = C.c_void_p(L.approx_args_init())
    L.approx_args_set_maxrank(aa, C.c_size_t(8)); L.approx_args_set_startrank(aa, C.c_size_t(3)); L.approx_args_set_kickrank(aa, C.c_size_t(2))
    L.approx_args_set_cross_tol(aa, C.c_double(1e-6)); L.approx_args_set_round_tol(aa, C.c_double(1e-6))
    vf = C.c_void_p(L.c3control_init_value(ctl.h, start, None, aa, 0))
    diag = C.c_void_p(None)
    v2 = C.c_void_p(L.c3control_pi_solve(ctl.h, C.c_size_t(2), C.c_double(1e-9), vf, aa, ctl.opt, 0, C.byref(diag)))
    v3 = C.c_void_p(L.c3control_vi_solve(ctl.h, C.c_size_t(2), C.c_double(1e-9), v2, aa, ctl.opt, 0, C.byref(diag)))
    for v in (vf, v2, v3): L.valuef_destroy(v)
    L.diag_destroy(C.byref(diag)); L.approx_args_free(aa)
    ctl.close()
    if it == 5: f1, r1 = free_mb(), rss_mb()
    if it % 10 == 0: print("facade", it, "free MB", round(free_mb(), 1), "rss MB", round(rss_mb(), 1), flush=True)
print("facade loop, iteration 5 -> end: device free", round(f1, 1), "->", round(free_mb(), 1), "MB; host max RSS", round(r1, 1), "->", round(rss_mb(), 1), "MB")
