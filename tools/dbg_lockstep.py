"""Lock-step comparison of the device loop and the oracle-fed loop of tests/regression_lib.py, update by update and, inside
the first update, sweep by sweep."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import regression_lib as R  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "pi_25"
if case == "car7d":
    from c3sc_amd import workloads as wl
    w0 = wl.c4_car7d().scaled(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4)
    cands = np.array([[a, b] for a in (-0.5, 0.07, 0.43) for b in (-1.0, 0.13, 0.91)])
    w = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, w0.ranks, w0.discount, w0.bc, list(w0.obstacles), cands)
    wts = np.array([0.3, 0.5, 0.2, 0.1, 0.15, 0.7, 0.25])
    case = dict(w=w, max_updates=21, conv=1e-9, adapt=1, startrank=3, maxrank=5, kick=2, cross_tol=1e-10, round_tol=1e-9,
                start_fn=lambda X: 1.0 + ((X - 0.1) ** 2 * wts).sum(axis=1))
gpu, orc = R.GpuLoop(case), R.OracleLoop(case)
L = gpu.L
state = gpu.init_value()
print("init rank", gpu.rank(state), "norm", gpu.norm(state))
for u in range(6):
    # inside the update: pi_solve then vi_solve, compared separately
    pa = gpu.pi_solve(10, gpu.conv, C.c_void_p(L.valuef_copy(state)))
    pb = orc.pi_solve(10, orc.conv, C.c_void_p(L.valuef_copy(state)))
    na, nb = gpu.nodal(pa), orc.nodal(pb)
    d = np.abs(na - nb)
    print(f"update {u}: after pi_solve rel diff {d.max() / np.abs(nb).max():.3e} at {np.unravel_index(d.argmax(), d.shape)} ranks {gpu.rank(pa)} {orc.rank(pb)}")
    va = gpu.vi_solve(1, gpu.conv, pb)
    vb = orc.vi_solve(1, orc.conv, pb)
    na, nb = gpu.nodal(va), orc.nodal(vb)
    d = np.abs(na - nb)
    print(f"          vi_solve from the oracle's pi result: rel diff {d.max() / np.abs(nb).max():.3e} ranks {gpu.rank(va)} {orc.rank(vb)}")
    # one policy-evaluation sweep only
    qa = gpu.pi_solve(1, gpu.conv, C.c_void_p(L.valuef_copy(state)))
    qb = orc.pi_solve(1, orc.conv, C.c_void_p(L.valuef_copy(state)))
    d = np.abs(gpu.nodal(qa) - orc.nodal(qb))
    print(f"          one pi sweep: rel diff {d.max() / np.abs(orc.nodal(qb)).max():.3e} at {np.unravel_index(d.argmax(), d.shape)}")
    state = va
