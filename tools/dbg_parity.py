import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine
w = wl.c4_car7d().scaled(ngrid=(11,) * 7, rank=10)
cores = wl.synth_cores(w)
P = oracle_lib.Problem(w, cores)
eng = BellmanEngine(0); eng.configure(w, cores)
eng.set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for k in range(7):
    idx = wl.synth_fibers(w, k, 37); idx[0, :] = 0; idx[1, :] = np.array(w.ngrid) - 1; idx[:, k] = 0
    ref, _, ab0 = P.bellman_fibers(k, idx)
    out, ui, ab = eng.bellman_fibers_host(k, idx)
    e = np.abs(out - ref)
    bad = np.argwhere(e > 1e-9 * np.abs(ref).max())
    print("k", k, eng.last_kernel(), "max err", e.max(), "nbad", len(bad), "bad fibers", sorted(set(bad[:, 0]))[:10], "bad nodes", sorted(set(bad[:, 1]))[:12])
    bab = np.argwhere(ab != ab0)
    for (f, j) in bab[:8]:
        print("   absorbed mismatch fiber", f, "node", j, "got", ab[f, j], "want", ab0[f, j], "idx", idx[f], "out", out[f, j], "ref", ref[f, j])
print("---- periodic moved to dim 3, dim 2 reflect")
w2 = wl.c4_car7d().scaled(ngrid=(11,) * 7, rank=10)
w2.bc = (1, 1, 3, 2, 3, 3, 3)
P2 = oracle_lib.Problem(w2, cores)
eng2 = BellmanEngine(0); eng2.configure(w2, cores)
for k in (2, 3):
    idx = wl.synth_fibers(w2, k, 37); idx[:, k] = 0
    ref, _, _ = P2.bellman_fibers(k, idx)
    out, ui, ab = eng2.bellman_fibers_host(k, idx)
    e = np.abs(out - ref)
    print("k", k, eng2.last_kernel(), "max err", e.max())
