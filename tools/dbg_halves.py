"""bellman_vi_batch_idx on a whole batch vs on its halves (separate value-iteration epochs), through libc3sc.so"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import facade_lib  # noqa: E402
from c3sc_amd import workloads as wl  # noqa: E402

L = facade_lib.lib()
w = wl.c4_car7d().scaled(ngrid=(11,) * 7, rank=4)
cores = wl.synth_cores(w)
ctl = facade_lib.Control(w)
vf = ctl.valuef(cores)
i32p = C.POINTER(C.c_int32)
for k in (0, 3, 6):
    idx = np.ascontiguousarray(wl.synth_fibers(w, k, 37), dtype=np.int32)
    N = w.ngrid[k]

    def run(sub):
        vi = ctl.begin_vi(vf)
        out = np.zeros((len(sub), N))
        rc = L.bellman_vi_batch_idx(C.c_size_t(len(sub)), C.c_size_t(k), sub.ctypes.data_as(i32p), facade_lib.dp(out), vi)
        assert rc == 0
        ctl.end_vi(vi)
        return out

    full = run(idx)
    a, b = run(np.ascontiguousarray(idx[:19])), run(np.ascontiguousarray(idx[19:]))
    print("k", k, "halves vs full max abs diff", np.abs(np.concatenate([a, b]) - full).max(), "scale", np.abs(full).max())
    # same epoch: first half then second half with ONE vi (memo shared)
    vi = ctl.begin_vi(vf)
    o1, o2 = np.zeros((19, N)), np.zeros((18, N))
    s1, s2 = np.ascontiguousarray(idx[:19]), np.ascontiguousarray(idx[19:])
    L.bellman_vi_batch_idx(C.c_size_t(19), C.c_size_t(k), s1.ctypes.data_as(i32p), facade_lib.dp(o1), vi)
    L.bellman_vi_batch_idx(C.c_size_t(18), C.c_size_t(k), s2.ctypes.data_as(i32p), facade_lib.dp(o2), vi)
    ctl.end_vi(vi)
    print("      same epoch, two calls:", np.abs(np.concatenate([o1, o2]) - full).max())
    # offset pointers into one buffer, as the sharded wrapper does
    vi = ctl.begin_vi(vf)
    out = np.zeros((37, N))
    base = out.ctypes.data
    L.bellman_vi_batch_idx.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bellman_vi_batch_idx(18, k, idx.ctypes.data + 19 * w.dx * 4, base + 19 * N * 8, vi)
    ctl.end_vi(vi)
    print("      second half through offset pointers:", np.abs(out[19:] - full[19:]).max(), "first half untouched:", np.abs(out[:19]).max())
    L.bellman_vi_batch_idx.argtypes = None
