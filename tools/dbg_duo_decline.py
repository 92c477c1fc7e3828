import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine
w = wl.WORKLOADS["quad10d"]().scaled(ngrid=(7, 6, 5, 8, 7, 6, 5, 8, 7, 25), rank=15)
cores = wl.synth_cores(w)
eng = BellmanEngine(0); eng.set_variant(4); eng.configure(w, cores)
for k in range(w.dx):
    idx = wl.synth_fibers(w, k, 70); idx[:, k] = 0
    out, ui, ab = eng.bellman_fibers_host(k, idx)
    print(k, eng.last_kernel(), flush=True)
