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
