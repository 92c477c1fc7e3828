"""Dense solution of the reference's Test_bellman_pi3d problem (tprob_test.c:2448-2540) -- every node of the 25^3 grid, no
function-train format, no cross approximation -- to show WHY the goal box the reference asserts (|x_i| < 0.4 at the end of the
noise-free closed loop, :2530-2535) is not reached: it is a property of the discrete stochastic problem itself (unit noise, absorbing
faces of cost 100 one unit away from the origin along x2, discount 0.1), not of the low-rank solver.

Method: policy iteration on the Markov chain of nodeutil.c:267-406 / bellman.c:88-112 restated with numpy (upwind rates, dt = h^2/Q,
discount exp(-beta dt)), 5^3 control candidates over [-5, 5]^3 (tprob3d workload), policy evaluation by a sparse direct solve.  The
result is then CHECKED by the oracle: one sweep of its bellman_vi over every fiber of the grid (the dense tensor as an exact
full-rank nodal train) must return the same tensor to 1e-9 -- i.e. V* is a fixed point of the oracle's operator.
Writes tests/golden/pi3d_dense_vstar.npz (V*, 125 KB).      python tools/run_reference_pi3d_dense.py      (CPU, ~1 minute)"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from c3sc_amd import workloads as wl  # noqa: E402


def exact_train(V):
    """nodal cores (reference layout cores[m][j][a + b r_m]) of the dense 3-D tensor at full rank: V[i,j,k] = sum_ab e_i[a] V[a,j,b] e_k[b]"""
    n0, n1, n2 = V.shape
    c0 = np.eye(n0)                                                                 # core 0: [i][b], r0 = 1
    c1 = np.ascontiguousarray(V.transpose(1, 2, 0)).reshape(n1, n2 * n0)            # core 1: [j][a + b n0] = V[a, j, b]
    c2 = np.eye(n2)                                                                 # core 2: [k][a], r3 = 1
    return (1, n0, n2, 1), [c0, c1, c2]


def oracle_sweep(w0, V):
    ranks, cores = exact_train(V)
    w = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, ranks, w0.discount, w0.bc, [], w0.cands)
    P = oracle_lib.Problem(w, cores, consistent_ends=True)
    n = w0.ngrid
    idx = np.array([[0, j, k] for j in range(n[1]) for k in range(n[2])], dtype=np.int32)
    out, ui, _ = P.bellman_fibers(0, idx, want_absorbed=False)
    return out.reshape(n[1], n[2], n[0]).transpose(2, 0, 1), ui.reshape(n[1], n[2], n[0]).transpose(2, 0, 1)


def main():
    w = wl.WORKLOADS["tprob3d"]()
    n, d = w.ngrid, 3
    xg = w.xgrid()
    h = [xg[m][1] - xg[m][0] for m in range(d)]
    h2 = min(h) ** 2
    t1 = [h2 / h[m] for m in range(d)]
    t2 = [h2 / h[m] / h[m] for m in range(d)]
    X = np.stack(np.meshgrid(*xg, indexing="ij"), axis=-1).reshape(-1, d)      # node -> coordinates
    nn = X.shape[0]
    ii = np.stack(np.meshgrid(*[np.arange(k) for k in n], indexing="ij"), axis=-1).reshape(-1, d)
    face = np.zeros(nn, dtype=bool)
    for m in range(d):
        face |= (ii[:, m] == 0) | (ii[:, m] == n[m] - 1)                        # every face absorbing (c3control_create's default)
    strides = [n[1] * n[2], n[2], 1]
    U = np.asarray(w.cands)
    nc = U.shape[0]
    # per candidate: rates p[m][-/+], dt, discount factor, stage (transition_assemble, bellmanrhs)
    rates = np.zeros((nc, nn, 2 * d))
    for c, u in enumerate(U):
        b = np.stack([X[:, 0] * X[:, 2] ** 2 * u[0], -X[:, 1] * u[2] + u[1], X[:, 0] * X[:, 1] * u[0] + 2 * u[1]], axis=1)  # f3, :223-251
        for m in range(d):
            half = t2[m] * 1.0 / 2.0                                              # sigma = I (s2, :197-220)
            rates[c, :, 2 * m] = half + np.where(b[:, m] < -1e-14, -t1[m] * b[:, m], 0.0)
            rates[c, :, 2 * m + 1] = half + np.where(b[:, m] > 1e-14, t1[m] * b[:, m], 0.0)
    Q = rates.sum(axis=2)
    dt = h2 / Q
    disc = np.exp(-w.discount * dt)
    stage_x = 0.2 * X[:, 0] ** 2 + 0.5 * X[:, 1] ** 2 + 2.0 * X[:, 2] ** 2       # stagecost3d, :273-300
    stage = stage_x[None, :] + (0.1 * U[:, 0] ** 2 + 0.5 * U[:, 1] ** 2 + 3.0 * U[:, 2] ** 2)[:, None]
    prob = rates / Q[:, :, None]
    nbr = np.zeros((nn, 2 * d), dtype=np.int64)
    for m in range(d):
        nbr[:, 2 * m] = np.arange(nn) - strides[m]
        nbr[:, 2 * m + 1] = np.arange(nn) + strides[m]
    nbr[face] = np.arange(nn)[face, None]                                        # absorbed nodes never use their neighbours
    interior = np.flatnonzero(~face)
    V = np.zeros(nn)
    V[face] = 100.0                                                              # boundcost, :302-309
    pol = np.zeros(nn, dtype=np.int64)
    t0 = time.time()
    for it in range(100):
        # improvement: greedy candidate (list order, first minimum)
        vals = dt * stage + disc * (prob * V[nbr][None, :, :]).sum(axis=2)       # (nc, nn)
        newpol = np.argmin(vals, axis=0)
        changed = int((newpol[interior] != pol[interior]).sum())
        pol = newpol
        # evaluation: (I - disc_pi P_pi) V = dt_pi stage_pi on interior nodes, V = 100 on the faces
        rows = np.repeat(interior, 2 * d)
        cols = nbr[interior].reshape(-1)
        data = -(disc[pol[interior], interior][:, None] * prob[pol[interior], interior, :]).reshape(-1)
        A = sp.csr_matrix((data, (rows, cols)), shape=(nn, nn)) + sp.identity(nn, format="csr")
        rhs = np.where(face, 100.0, dt[pol, np.arange(nn)] * stage[pol, np.arange(nn)])
        V = spla.spsolve(A.tocsc(), rhs)
        print(f"policy iteration {it}: {changed} interior nodes changed their candidate, max V {V.max():.4f}, V at the start state ... {time.time() - t0:.1f} s", flush=True)
        if changed == 0 and it > 0:
            break
    Vd = V.reshape(n)
    Tv, ui = oracle_sweep(w, Vd)
    err = np.abs(Tv - Vd).max()
    print(f"oracle check: |T(V*) - V*|_max = {err:.3e} (max V* {Vd.max():.3f}); greedy candidates equal to the policy at "
          f"{float((ui.reshape(-1)[interior] == pol[interior]).mean()):.4f} of the interior nodes")
    assert err < 1e-8, err
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pi3d_dense_vstar.npz"), V=Vd, cands=U, oracle_fixed_point_error=err)
    i0, i1 = int(np.argmin(np.abs(xg[0]))), int(np.argmin(np.abs(xg[1])))
    print("profile of V* along x2 at the nodes nearest x0 = x1 = 0:")
    for k in range(n[2]):
        print(f"  x2 = {xg[2][k]:+.4f}  V* = {Vd[i0, i1, k]:.4f}")


if __name__ == "__main__":
    main()
