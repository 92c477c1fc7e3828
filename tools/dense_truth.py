"""Dense ground truth for the 7-D car on a reduced grid, and the error of the TT solver against it.

The TT solver's value iteration on car7d reaches a noise floor instead of a tolerance (bench.py: vi_iters_to_tol).  To tell
approximation error from solver defects, this tool solves the SAME discrete problem without any low-rank approximation: the
full tensor V on every node of a reduced grid (n^7 nodes, n = 9 ... 13), plain value iteration V <- T(V) with the Markov-chain
backup of nodeutil.c:267-406 / bellman.c:88-112 restated densely in torch (a third implementation beside the HIP kernels and
the C oracle; tests/test_dense_truth.py holds it to the oracle on a small grid).  End points follow the solver's consistent rule
(c3control_set_consistent_ends): a node on an absorbing face is absorbed whatever the direction.

Then, for a list of rank caps: the library's own value iteration (c3control_step_vi through libc3sc.so, device-resident cross
iterations) on the same grid from the same start, and per rank cap
  * the best the format can do: TT-SVD truncation error of V* at that rank,
  * the solver's distance to V* along its sweeps (relative L2 and max norm over all nodes),
  * its step size |V_i+1 - V_i| / |V| at the end (the "noise floor").

    python tools/dense_truth.py [n=11] [ranks=5,8,10] [tt_sweeps=1500] [dense_tol=1e-9]      (GPU box; torch on cuda:0)
"""
import ctypes as C
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from c3sc_amd import workloads as wl  # noqa: E402


# ------------------------------------------------------------------------------------------------ dense Bellman operator
class DenseCar7D:
    """T(V) for the car7d workload on the full grid (torch tensors of shape w.ngrid)."""

    def __init__(self, w, device):
        import torch

        assert w.model == wl.MODEL_CAR7D
        self.w, self.t, self.dev = w, torch, device
        d = w.dx
        xg = w.xgrid()
        self.N = list(w.ngrid)
        f64 = dict(dtype=torch.float64, device=device)

        def axis(m, vals):
            shp = [1] * d
            shp[m] = self.N[m]
            return torch.as_tensor(np.asarray(vals), **f64).reshape(shp)

        X = [axis(m, xg[m]) for m in range(d)]
        h = [xg[m][1] - xg[m][0] for m in range(d)]
        hmin = min(h)
        self.h2 = hmin * hmin                                   # mca_add_grid_refs, bellman.c:171-188
        self.tv = [(self.h2 / h[m], self.h2 / h[m] / h[m]) for m in range(d)]
        # drift of the control-independent dimensions (models.hpp: Car7D; host tables are libm cos / sin / tan)
        v, om = X[3], X[4]
        self.b_fixed = [v * axis(2, np.cos(xg[2])), v * axis(2, np.sin(xg[2])), om + 0 * v, 2.0 * X[6] + 0 * v,
                        (axis(3, xg[3] / (0.2 * (1.0 + xg[3] / 8.0))) * axis(5, np.tan(xg[5])) - om) / 0.5]
        self.sig = [1.0, 1.0] + [1e-2] * 5
        self.stage = 1.0 + X[0] * X[0] + X[1] * X[1]
        # flags: absorbing faces (boundcost 10) override obstacles (obscost 0): process_fibers_neighbor, consistent end points
        absorbed = torch.zeros(self.N, dtype=torch.bool, device=device)
        for m in range(d):
            if w.bc[m] == wl.BC_ABSORB:
                idx = [slice(None)] * d
                idx[m] = 0
                absorbed[tuple(idx)] = True
                idx[m] = self.N[m] - 1
                absorbed[tuple(idx)] = True
        inobs = torch.zeros(self.N, dtype=torch.bool, device=device)
        for cen, wid in w.obstacles:
            box = torch.ones(self.N, dtype=torch.bool, device=device)
            for m in range(d):
                lo, hi = cen[m] - wid[m] / 2.0, cen[m] + wid[m] / 2.0  # boundary.c:264-267, inclusive (:329-344)
                box = box & axis(m, (xg[m] >= lo) & (xg[m] <= hi)).to(torch.bool)
            inobs = inobs | box
        self.absorbed, self.inobs = absorbed, inobs & ~absorbed
        self.nb = []  # neighbour index vectors per dim (nodeutil.c:515-612; periodic: node 0 == node N-1, Q8)
        for m in range(d):
            n = self.N[m]
            lo, hi = np.arange(n) - 1, np.arange(n) + 1
            if w.bc[m] == wl.BC_PERIODIC:
                lo[0], hi[0], lo[n - 1], hi[n - 1] = n - 2, 1, n - 2, 1
            else:  # reflect (absorbing faces never use their neighbours)
                lo[0], hi[0], lo[n - 1], hi[n - 1] = 0, 1, n - 2, n - 1
            self.nb.append((torch.as_tensor(lo, device=device), torch.as_tensor(hi, device=device)))
        self.cands = [tuple(c) for c in w.cands]

    def apply(self, V):
        t, d = self.t, self.w.dx
        lo = [t.index_select(V, m, self.nb[m][0]) for m in range(d)]
        hi = [t.index_select(V, m, self.nb[m][1]) for m in range(d)]
        # rates of the control-independent dimensions, once
        Qf, Sf = None, None
        for m in range(5):
            t1, t2 = self.tv[m]
            half = t2 * self.sig[m] * self.sig[m] / 2.0
            b = self.b_fixed[m]
            pm = t.where(b < -1e-14, half - t1 * b, t.full_like(b, half))
            pp = t.where(b > 1e-14, half + t1 * b, t.full_like(b, half))
            q, s = pm + pp, pm * lo[m] + pp * hi[m]
            Qf, Sf = (q, s) if Qf is None else (Qf + q, Sf + s)
        best = None
        for (u0, u1) in self.cands:  # list order, strict '<' (first minimum)
            Q, S = Qf, Sf
            for m, b in ((5, u0), (6, u1)):
                t1, t2 = self.tv[m]
                half = t2 * self.sig[m] * self.sig[m] / 2.0
                pm = half - t1 * b if b < -1e-14 else half
                pp = half + t1 * b if b > 1e-14 else half
                Q = Q + (pm + pp)
                S = S + (pm * lo[m] + pp * hi[m])
            dt = self.h2 / Q
            # bellmanrhs (bellman.c:88-112): dt stage + exp(-beta dt) sum_i p_i V_i with p_i = rate_i / Q; the self transition
            # probability 1 - sum_i p_i is rounding noise (nodeutil.c:369-398) and is dropped here
            val = dt * self.stage + (S / Q if self.w.discount == 0.0 else t.exp(-self.w.discount * dt) * (S / Q))
            best = val if best is None else t.minimum(best, val)
        out = t.where(self.absorbed, t.full_like(best, 10.0), best)
        return t.where(self.inobs, t.zeros_like(out), out)


def tt_svd_error(V, r):
    """Relative Frobenius / max error of the rank-r TT-SVD truncation of the dense tensor V (numpy)."""
    shp, d = V.shape, V.ndim
    A, r0, cores = V.copy(), 1, []
    for m in range(d - 1):
        A = A.reshape(r0 * shp[m], -1)
        U, S, Vt = np.linalg.svd(A, full_matrices=False)
        rr = min(r, len(S))
        cores.append(U[:, :rr].reshape(r0, shp[m], rr))
        A, r0 = S[:rr, None] * Vt[:rr], rr
    cores.append(A.reshape(r0, shp[-1], 1))
    acc = cores[0]
    for G in cores[1:]:
        acc = np.tensordot(acc, G, axes=([acc.ndim - 1], [0]))
    rec = acc.reshape(shp)
    return float(np.linalg.norm(rec - V) / np.linalg.norm(V)), float(np.abs(rec - V).max() / np.abs(V).max())


def tt_to_dense(ngrid, ranks, cores, torch, device):
    acc = torch.ones((1, 1), dtype=torch.float64, device=device)
    for m in range(len(ngrid)):
        G = torch.as_tensor(cores[m], dtype=torch.float64, device=device).reshape(ngrid[m], ranks[m + 1], ranks[m]).permute(2, 0, 1)
        acc = torch.tensordot(acc, G, dims=([acc.ndim - 1], [0]))
    return acc.reshape(ngrid)


_VSTAR = {}


def reduced_grid_truth(n=9, rcap=9, tt_sweeps=600, dense_tol=1e-9, log=None, crossrank=0):
    """Dense V* on n^7 nodes, best rank-rcap error, and the library's value iteration at that rank cap: a dict for bench.py."""
    import torch

    import facade_lib

    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    w = wl.c4_car7d().scaled(ngrid=(n,) * 7, rank=4)
    if (n, dense_tol) in _VSTAR:
        V, it = _VSTAR[(n, dense_tol)]
    else:
        op = DenseCar7D(w, dev)
        V = torch.zeros(w.ngrid, dtype=torch.float64, device=dev)
        for it in range(100000):
            Vn = op.apply(V)
            step = float((Vn - V).abs().max())
            V = Vn
            if step < dense_tol:
                break
        _VSTAR[(n, dense_tol)] = (V, it)
    vnorm, vmax = float(V.norm()), float(V.abs().max())
    best2, bestm = tt_svd_error(V.cpu().numpy(), rcap)
    L = facade_lib.lib()
    for f in ("c3control_init_value", "c3control_step_vi"):
        getattr(L, f).restype = C.c_void_p
    for f in ("valuef_norm", "valuef_norm2diff"):
        getattr(L, f).restype = C.c_double
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
    zero = facade_lib.FIBER_FN(lambda N, x, out, a: (np.ctypeslib.as_array(out, shape=(N,)).fill(0.0), 0)[1])
    ctl = facade_lib.Control(w, consistent_ends=None)
    aa = C.c_void_p(L.approx_args_init())
    L.approx_args_set_cross_tol(aa, C.c_double(1e-6))
    L.approx_args_set_round_tol(aa, C.c_double(1e-6))
    L.approx_args_set_kickrank(aa, C.c_size_t(2))
    L.approx_args_set_startrank(aa, C.c_size_t(4))
    L.approx_args_set_maxrank(aa, C.c_size_t(rcap))
    L.approx_args_set_crossrank(aa, C.c_size_t(crossrank))
    v = C.c_void_p(L.c3control_init_value(ctl.h, zero, None, aa, 0))
    ne = C.c_size_t(0)
    errs, steps = [], []
    for ii in range(tt_sweeps):
        nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))
        steps.append(L.valuef_norm2diff(v, nxt) / max(L.valuef_norm(nxt), 1e-300))
        L.valuef_destroy(v)
        v = nxt
        if ii >= tt_sweeps // 2 and (ii % 25 == 0 or ii == tt_sweeps - 1):
            ranks = [int(L.valuef_get_ranks(v)[i]) for i in range(8)]
            pp = L.valuef_get_cores(v)
            cores = [np.ctypeslib.as_array(pp[m], shape=(n * ranks[m] * ranks[m + 1],)).copy() for m in range(7)]
            D = tt_to_dense(w.ngrid, ranks, cores, torch, dev)
            errs.append((float((D - V).norm()) / vnorm, float((D - V).abs().max()) / vmax))
    L.valuef_destroy(v)
    L.approx_args_free(aa)
    ctl.close()
    e2 = np.array([e[0] for e in errs])
    return {"grid": f"{n}^7", "nodes": int(V.numel()), "dense_vi_sweeps_to_1e-9": it + 1, "rank_cap": rcap, "cross_rank": crossrank if crossrank else rcap,
            "best_rank_cap_train_rel_L2": best2, "best_rank_cap_train_rel_max": bestm,
            "tt_vi_sweeps": tt_sweeps, "tt_vi_rel_L2_error_vs_dense_median": float(np.median(e2)), "tt_vi_rel_L2_error_vs_dense_min": float(e2.min()),
            "tt_vi_rel_L2_error_vs_dense_max": float(e2.max()), "tt_vi_rel_max_error_vs_dense_median": float(np.median([e[1] for e in errs])),
            "tt_vi_median_rel_step_second_half": float(np.median(steps[tt_sweeps // 2:])),
            "what": "car7d on a reduced grid: V* by dense value iteration (every node, no low-rank format; tools/dense_truth.py), the error of its "
                    "best rank-capped train (TT-SVD), and the library's value iteration at that cap measured against V* over the second half of "
                    "its sweeps: the solver sits a small multiple above the format's own error, and its step noise is below that error"}


def main():
    import torch

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    rank_caps = [int(r) for r in (sys.argv[2] if len(sys.argv) > 2 else "5,8,10").split(",")]
    tt_sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    dense_tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-9
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    w = wl.c4_car7d().scaled(ngrid=(n,) * 7, rank=4)
    op = DenseCar7D(w, dev)
    V = torch.zeros(w.ngrid, dtype=torch.float64, device=dev)
    t0 = time.time()
    for it in range(200000):
        Vn = op.apply(V)
        step = float((Vn - V).abs().max())
        V = Vn
        if it % 500 == 0:
            print(f"dense VI sweep {it:6d}: max step {step:.3e}  max V {float(V.max()):.6f}  {time.time() - t0:.1f} s", flush=True)
        if step < dense_tol:
            break
    vnorm, vmax = float(V.norm()), float(V.abs().max())
    print(f"dense value iteration on {n}^7 = {V.numel()} nodes: {it + 1} sweeps to max step < {dense_tol:g} in {time.time() - t0:.1f} s; "
          f"|V*|_2 = {vnorm:.6f}, max = {vmax:.6f}", flush=True)
    Vh = V.cpu().numpy()
    for r in sorted(set(rank_caps + [15, 20])):
        if r <= n:
            ef, em = tt_svd_error(Vh, r)
            print(f"best rank-{r:2d} train (TT-SVD of V*): relative L2 error {ef:.3e}, max-norm error {em:.3e}", flush=True)

    # the library's value iteration at each rank cap
    import facade_lib

    L = facade_lib.lib()
    for f in ("c3control_init_value", "c3control_step_vi"):
        getattr(L, f).restype = C.c_void_p
    for f in ("valuef_norm", "valuef_norm2diff"):
        getattr(L, f).restype = C.c_double
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
    zero = facade_lib.FIBER_FN(lambda N, x, out, a: (np.ctypeslib.as_array(out, shape=(N,)).fill(0.0), 0)[1])
    for rcap in rank_caps:
        ctl = facade_lib.Control(w, consistent_ends=None)
        aa = C.c_void_p(L.approx_args_init())
        L.approx_args_set_cross_tol(aa, C.c_double(1e-6))
        L.approx_args_set_round_tol(aa, C.c_double(1e-6))
        L.approx_args_set_kickrank(aa, C.c_size_t(2))
        L.approx_args_set_startrank(aa, C.c_size_t(4))
        L.approx_args_set_maxrank(aa, C.c_size_t(rcap))
        v = C.c_void_p(L.c3control_init_value(ctl.h, zero, None, aa, 0))
        ne = C.c_size_t(0)
        t1 = time.time()
        best = (1e300, -1)
        for ii in range(tt_sweeps):
            nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))
            step = L.valuef_norm2diff(v, nxt) / max(L.valuef_norm(nxt), 1e-300)
            L.valuef_destroy(v)
            v = nxt
            if ii % max(1, tt_sweeps // 15) == 0 or ii == tt_sweeps - 1:
                ranks = [int(L.valuef_get_ranks(v)[i]) for i in range(8)]
                pp = L.valuef_get_cores(v)
                cores = [np.ctypeslib.as_array(pp[m], shape=(n * ranks[m] * ranks[m + 1],)).copy() for m in range(7)]
                D = tt_to_dense(w.ngrid, ranks, cores, torch, dev)
                e2, em = float((D - V).norm()) / vnorm, float((D - V).abs().max()) / vmax
                best = min(best, (e2, ii))
                print(f"rank cap {rcap:2d} sweep {ii:5d}: relative step {step:.3e}  error vs V*: L2 {e2:.3e}  max {em:.3e}  ranks {ranks[1:-1]}", flush=True)
        print(f"rank cap {rcap:2d}: {tt_sweeps} sweeps in {time.time() - t1:.1f} s; final relative step {step:.3e}; final error vs V* L2 {e2:.3e} max {em:.3e}; "
              f"smallest L2 error along the way {best[0]:.3e} (sweep {best[1]})", flush=True)
        L.valuef_destroy(v)
        L.approx_args_free(aa)
        ctl.close()


if __name__ == "__main__":
    main()
