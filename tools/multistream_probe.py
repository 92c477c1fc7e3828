"""Per-rank load of an 8-GPU strong-scaling step (car7d, 2^17 fibers per varying dimension, seven launches): the launches are
independent, so they may be issued on several streams; this times a step with 1, 2, 3 and 7 streams.
    python tools/multistream_probe.py [fibers]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3sc_amd import workloads as wl
from c3sc_amd.engine import BellmanEngine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
w = wl.c4_car7d(); cores = wl.synth_cores(w)
eng = BellmanEngine(0); eng.configure(w, cores)
d = w.dx
idx = [torch.from_numpy(wl.synth_fibers(w, k, F)).cuda() for k in range(d)]
out = [torch.empty((F, w.ngrid[k]), dtype=torch.float64, device="cuda") for k in range(d)]
for ns in (1, 2, 3, 7):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for rep in range(3):
        for k in range(d):
            eng.bellman_fibers(k, idx[k], out[k], stream_ptr=streams[k % ns].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for rep in range(R):
        for k in range(d):
            eng.bellman_fibers(k, idx[k], out[k], stream_ptr=streams[k % ns].cuda_stream)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / R
    print(f"{ns} stream(s): {ms:.3f} ms per step of {d} launches of {F} fibers ({ms / d:.4f} ms per launch)")
