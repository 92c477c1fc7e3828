"""Parity of one workload at a list of ranks (incl. ranks that are padded to the next compiled RP), per kernel variant.
    python tools/dbg_rank_pad.py [workload] [ngrid] [ranks...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine
name = sys.argv[1] if len(sys.argv) > 1 else "lqg2d"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 19
ranks = [int(a) for a in sys.argv[3:]] or [3, 4, 5, 8, 9, 12, 13, 16, 17, 20]
w0 = wl.WORKLOADS[name]()
for r in ranks:
    w = w0.scaled(ngrid=(n,) * w0.dx, rank=r)
    cores = wl.synth_cores(w)
    P = oracle_lib.Problem(w, cores)
    eng = BellmanEngine(0); eng.configure(w, cores)
    for variant in (0, 1, 3):
        try:
            eng.set_variant(variant)
            for k in range(w.dx):
                idx = wl.synth_fibers(w, k, 37); idx[:, k] = 0
                ref, _, _ = P.bellman_fibers(k, idx)
                out, ui, ab = eng.bellman_fibers_host(k, idx)
                print(f"rank {r:2d} variant {variant} k {k} {eng.last_kernel():60s} max rel err {np.abs(out - ref).max() / np.abs(ref).max():.3e}", flush=True)
        except Exception as e:  # no instantiation for this (model, rank, variant)
            print(f"rank {r:2d} variant {variant}: {str(e)[:100]}")
