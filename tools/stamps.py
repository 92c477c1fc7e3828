import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["C3SC_DBG"] = os.environ.get("C3SC_DBG", "128")
import torch
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine
w = wl.c4_car7d(); cores = wl.synth_cores(w)
eng = BellmanEngine(0); eng.configure(w, cores); eng.set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
F = 1 << 17
names = ["setup", "fold-compute", "swap", "partials+wr", "barrier1", "fin-rest", "barrier2", "fold-stage", "fin-pre", "fin-scan", "x", "y"]
for k in (0, 3, 6):
    idx = torch.from_numpy(wl.synth_fibers(w, k, F)).cuda()
    out = eng.bellman_fibers(k, idx); torch.cuda.synchronize()
    out = eng.bellman_fibers(k, idx); torch.cuda.synchronize()
    var = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    if var == 3:
        d = eng.debug_read(2048 * 12).reshape(2048, 12).astype(np.float64)
        for h in (0, 1):
            m = d[h::2].mean(axis=0)
            print(f"k={k} wave{h}: total {m.sum():.0f} cyc | " + " ".join(f"{n}={v:.0f}" for n, v in zip(names, m)))
    else:
        d = eng.debug_read(1024 * 8).reshape(1024, 8).astype(np.float64)
        m = d.mean(axis=0)
        print(f"k={k}: total {m.sum():.0f} cyc | setup={m[0]:.0f} fold={m[1]:.0f} FT={m[2]:.0f} backup={m[3]:.0f}")
