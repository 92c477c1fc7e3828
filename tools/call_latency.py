"""Latency of one small batch through the host-buffer entry point (what a cross-approximation core step pays):
total per call, and its parts -- device launch + sync alone, torch H2D / D2H copies of the same sizes, status read.
    python tools/call_latency.py [workload] [fibers]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine
name = sys.argv[1] if len(sys.argv) > 1 else "car7d"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w = wl.WORKLOADS[name]()
eng = BellmanEngine(0); eng.configure(w, wl.synth_cores(w))
k = 3 % w.dx
idx = wl.synth_fibers(w, k, F)
reps = 2000
def timeit(fn, reps=reps):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
t_host = timeit(lambda: eng.bellman_fibers_host(k, idx, want_uidx=False, want_absorbed=False))
idx_t = torch.from_numpy(idx).cuda(); out_t = torch.empty((F, w.ngrid[k]), dtype=torch.float64, device="cuda")
def dev():
    eng.bellman_fibers(k, idx_t, out_t); torch.cuda.synchronize()
t_dev = timeit(dev)
eng.timer_start(torch.cuda.current_stream().cuda_stream)
for _ in range(200): eng.bellman_fibers(k, idx_t, out_t)
t_kernel = eng.timer_stop(torch.cuda.current_stream().cuda_stream) / 200 * 1e3
t_status = timeit(lambda: eng.status())
h_out = np.empty((F, w.ngrid[k]))
import ctypes as C
hip = C.CDLL("libamdhip64.so")
def h2d(): hip.hipMemcpy(C.c_void_p(idx_t.data_ptr()), C.c_void_p(idx.ctypes.data), C.c_size_t(idx.nbytes), C.c_int(1))
def d2h(): hip.hipMemcpy(C.c_void_p(h_out.ctypes.data), C.c_void_p(out_t.data_ptr()), C.c_size_t(h_out.nbytes), C.c_int(2))
t_h2d, t_d2h = timeit(h2d), timeit(d2h)
print(f"{name} F={F} ({eng.last_kernel()}): host-buffer call {t_host:.1f} us | launch+sync {t_dev:.1f} us (kernel {t_kernel:.1f} us back-to-back) | "
      f"hipMemcpy H2D {idx.nbytes} B {t_h2d:.1f} us, D2H {h_out.nbytes} B {t_d2h:.1f} us | status read {t_status:.1f} us")
