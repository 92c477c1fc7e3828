"""Host-buffer entry point (c3sc_hip_bellman_fibers_host: pageable numpy arrays in and out) at the bench size:
the PCIe-inclusive rate next to the HBM-resident one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine
w = wl.c4_car7d(); cores = wl.synth_cores(w)
eng = BellmanEngine(0); eng.configure(w, cores); eng.set_variant(3)
F = 1 << 17
for k in (0, 3):
    idx = wl.synth_fibers(w, k, F)
    eng.bellman_fibers_host(k, idx, want_uidx=False, want_absorbed=False)
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        eng.bellman_fibers_host(k, idx, want_uidx=False, want_absorbed=False)
    dt = (time.perf_counter() - t0) / n
    print(f"k={k}: host-buffer call {dt*1e3:.2f} ms per launch of {F} fibers -> {F*w.ngrid[k]/dt:.3e} nodes/s (in: {idx.nbytes/1e6:.1f} MB, out: {F*w.ngrid[k]*8/1e6:.1f} MB)")
