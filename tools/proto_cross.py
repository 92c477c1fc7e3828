"""CPU prototype of the cross-approximation step of the value iteration (numpy + the oracle's fibers), used to size algorithmic
changes to c3sc_cross.c / cross_device.hip before they are written in C and HIP:

    python tools/proto_cross.py [n=9] [rank=9] [sweeps=300] [variant=interp|ls] [extra=rank]

car7d on the reduced n^7 grid: dense V* (tools/dense_truth.py's operator on the CPU, cached under /tmp), then TT value iteration
with the chosen core step; prints the relative step and the error against V* along the sweeps."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_lib  # noqa: E402
from c3sc_amd import workloads as wl  # noqa: E402


def dense_vstar(n, tol=1e-9):
    path = f"/tmp/c3sc_vstar_car7d_{n}.npy"
    if os.path.exists(path):
        return np.load(path)
    import torch

    import dense_truth

    torch.set_num_threads(8)
    w = wl.c4_car7d().scaled(ngrid=(n,) * 7, rank=4)
    op = dense_truth.DenseCar7D(w, torch.device("cpu"))
    V = torch.zeros(w.ngrid, dtype=torch.float64)
    t0 = time.time()
    for it in range(100000):
        Vn = op.apply(V)
        step = float((Vn - V).abs().max())
        V = Vn
        if it % 50 == 0:
            print(f"dense sweep {it} step {step:.3e} {time.time() - t0:.0f}s", flush=True)
        if step < tol:
            break
    np.save(path, V.numpy())
    return V.numpy()


def tt_dense(cores):
    """cores[k]: (r0, N, r1)"""
    acc = cores[0].reshape(cores[0].shape[1], cores[0].shape[2])
    for G in cores[1:]:
        acc = np.tensordot(acc, G, axes=([acc.ndim - 1], [0]))
    return acc[..., 0]


def tt_dot(a, b):
    M = np.ones((1, 1))
    for A, B in zip(a, b):
        # M[ra, rb]; A (ra, N, ra'), B (rb, N, rb')
        P = np.tensordot(M, B, axes=([1], [0]))  # (ra, N, rb')
        M = np.tensordot(A, P, axes=([0, 1], [0, 1]))  # (ra', rb')
    return float(M[0, 0])


def maxvol(A, rows0=None, tol=1.05, maxit=500):
    """rows of (quasi-)maximal volume of the tall A (m x r); returns rows, B = A inv(A[rows])."""
    m, r = A.shape
    A = A + 1e-10 * (np.abs(A).max() + 1e-300) * np.random.default_rng(m * 131 + r).standard_normal(A.shape)  # rank-deficient starts (T(0))
    rows = None
    if rows0 is not None and len(rows0) == r and len(set(rows0)) == r:
        sub = A[rows0]
        if np.linalg.cond(sub) < 1e13:
            rows = list(rows0)
    if rows is None:
        import scipy.linalg as sla

        _, _, piv = sla.qr(A.T, pivoting=True, mode="economic")
        rows = list(piv[:r])
    B = np.linalg.solve(A[rows].T, A.T).T
    nsw = 0
    for _ in range(maxit):
        i, j = np.unravel_index(np.argmax(np.abs(B)), B.shape)
        if abs(B[i, j]) <= tol:
            break
        rows[j] = i
        B = B - np.outer(B[:, j], (B[i, :] - np.eye(r)[j])) / B[i, j]
        nsw += 1
    return rows, B, nsw


def rect_extra(B, rows, p, cand=None):
    """greedy rectangular maxvol: p more rows (from cand, default all) maximising the residual row norm of B pinv(B[chosen])"""
    chosen = list(rows)
    C = B.copy()  # C = A pinv(A[chosen]) restricted to the column space: start with B (B[rows] = I)
    Acur = B
    mask = np.ones(B.shape[0], dtype=bool)
    mask[chosen] = False
    if cand is not None:
        m2 = np.zeros_like(mask)
        m2[cand] = True
        mask &= m2
    for _ in range(p):
        if not mask.any():
            break
        Cc = Acur @ np.linalg.pinv(Acur[chosen])
        nr = (Cc * Cc).sum(axis=1)
        nr[~mask] = -1.0
        i = int(np.argmax(nr))
        chosen.append(i)
        mask[i] = False
    return chosen[len(rows):]


def tt_round_rank(cores, rmax):
    """TT-SVD rounding to ranks <= rmax: right-to-left orthogonalisation, then left-to-right truncated SVDs"""
    cores = [c.copy() for c in cores]
    d = len(cores)
    for k in range(d - 1, 0, -1):
        r0, N, r1 = cores[k].shape
        Q, R = np.linalg.qr(cores[k].reshape(r0, N * r1).T)  # (N r1, q), (q, r0)
        q = Q.shape[1]
        cores[k] = Q.T.reshape(q, N, r1)
        cores[k - 1] = np.tensordot(cores[k - 1], R.T, axes=([2], [0]))
    for k in range(d - 1):
        r0, N, r1 = cores[k].shape
        U, S, Vt = np.linalg.svd(cores[k].reshape(r0 * N, r1), full_matrices=False)
        rr = min(rmax, len(S))
        cores[k] = U[:, :rr].reshape(r0, N, rr)
        cores[k + 1] = np.tensordot(S[:rr, None] * Vt[:rr], cores[k + 1], axes=([1], [0]))
    return cores


class Cross:
    def __init__(self, w, r, variant="interp", extra=0, swap_tol=0.05, consistent=True, fit="ls"):
        self.w, self.d, self.N = w, w.dx, list(w.ngrid)
        self.r = [1] + [r] * (self.d - 1) + [1]
        for k in range(1, self.d):  # ranks cannot exceed the size of either unfolding side
            self.r[k] = min(self.r[k], int(np.prod(self.N[:k])), int(np.prod(self.N[k:])))
        self.variant, self.p, self.tol = variant, extra, 1.0 + swap_tol
        self.fit = fit
        self.P = oracle_lib.Problem(w, consistent_ends=consistent)
        d, N = self.d, self.N
        rng = np.random.default_rng(7)
        # I[k]: left tuples over dims < k (first r[k] = core, rest = extras); J[k]: right tuples over dims > k
        self.I = [np.zeros((1, 0), dtype=np.int32)] + [self._rand_tuples(rng, k, self.r[k]) for k in range(1, d)]
        self.J = [self._rand_tuples_r(rng, k, self.r[k + 1]) for k in range(d - 1)] + [np.zeros((1, 0), dtype=np.int32)]
        self.Iext = [np.zeros((0, k), dtype=np.int32) for k in range(d)]
        self.nfib = 0
        self.nswaps = 0
        self.frozen = False

    def _rand_tuples(self, rng, k, n):
        return np.stack([rng.integers(0, self.N[m], size=n) for m in range(k)], axis=1).astype(np.int32)

    def _rand_tuples_r(self, rng, k, n):
        return np.stack([rng.integers(0, self.N[m], size=n) for m in range(k + 1, self.d)], axis=1).astype(np.int32)

    def set_value(self, cores):
        ranks = [1] + [c.shape[2] for c in cores]
        ref = [np.ascontiguousarray(np.transpose(c, (1, 2, 0)).reshape(c.shape[1], -1)) for c in cores]  # [j][b][a] -> a + b r0
        self.vf = oracle_lib.ValueF(self.N, ranks, ref)
        self.P.L.orc_problem_set_value(self.P.h, self.vf.h)

    def fibers(self, k, L, R):
        """values T(L[a], :, R[b]) -> (len L, N, len R)"""
        nl, nr = len(L), len(R)
        idx = np.zeros((nl * nr, self.d), dtype=np.int32)
        idx[:, :k] = np.repeat(L, nr, axis=0)
        idx[:, k + 1:] = np.tile(R, (nl, 1))
        out, _, _ = self.P.bellman_fibers(k, idx, want_absorbed=False)
        self.nfib += nl * nr
        return out.reshape(nl, nr, self.N[k]).transpose(0, 2, 1)

    def sweep_lr(self):
        d, N = self.d, self.N
        changed = False
        for k in range(d - 1):
            r0, r1 = self.r[k], self.r[k + 1]
            L = self.I[k]
            C = self.fibers(k, L, self.J[k]).reshape(r0 * N[k], -1)  # rows (a, j): a * N + j
            old = self.I[k + 1]
            rows0 = []
            for q in range(len(old)):
                hit = [a for a in range(r0) if k == 0 or np.array_equal(L[a], old[q][:k])]
                if hit:
                    rows0.append(hit[0] * N[k] + int(old[q][k]))
            rows, B, ns = maxvol(C, rows0 if len(rows0) == r1 else None, 1e300 if self.frozen else self.tol)
            self.nswaps += ns
            rows_sorted = sorted(rows)
            new = np.array([list(L[rw // N[k]]) + [rw % N[k]] for rw in rows_sorted], dtype=np.int32).reshape(r1, k + 1)
            if not np.array_equal(new, old):
                changed = True
            self.I[k + 1] = new
            sticky = os.environ.get("STICKY") == "1" and len(self.Iext[k + 1]) == self.p and np.array_equal(new, old)
            if self.p > 0 and not self.frozen and not sticky:
                ex = rect_extra(B, rows, self.p)
                self.Iext[k + 1] = np.array([list(L[rw // N[k]]) + [rw % N[k]] for rw in sorted(ex)], dtype=np.int32).reshape(len(ex), k + 1)
        return changed

    def sweep_rl(self):
        d, N = self.d, self.N
        cores = [None] * d
        changed = False
        for k in range(d - 1, 0, -1):
            r0, r1 = self.r[k], self.r[k + 1]
            R = self.J[k]
            Lall = np.concatenate([self.I[k], self.Iext[k]], axis=0) if self.p > 0 else self.I[k]
            Cfull = self.fibers(k, Lall, R)  # (K, N, r1)
            K = Cfull.shape[0]
            Ct_full = Cfull.reshape(K, N[k] * r1).T  # rows (j, b): j * r1 + b ; cols: left tuples
            Ct = Ct_full[:, :r0]
            old = self.J[k - 1]
            rows0 = []
            for q in range(len(old)):
                hit = [b for b in range(r1) if k == d - 1 or np.array_equal(R[b], old[q][1:])]
                if hit:
                    rows0.append(int(old[q][0]) * r1 + hit[0])
            tol = 1e300 if self.frozen else self.tol
            if self.p > 0 and self.fit in ("svd", "svdls"):
                U, S, Vt = np.linalg.svd(Ct_full, full_matrices=False)
                rows, B, ns = maxvol(U[:, :r0], rows0 if len(rows0) == r0 else None, tol)
            else:
                rows, B, ns = maxvol(Ct, rows0 if len(rows0) == r0 else None, tol)
            self.nswaps += ns
            order = np.argsort(rows)
            rows = [rows[i] for i in order]
            B = B[:, order]
            if self.p > 0 and self.fit in ("ls", "svdls"):
                B = Ct_full @ np.linalg.pinv(Ct_full[rows])  # least squares over the K left tuples
            new = np.array([[rw // r1] + list(R[rw % r1]) for rw in rows], dtype=np.int32).reshape(r0, d - k)
            if not np.array_equal(new, old):
                changed = True
            self.J[k - 1] = new
            cores[k] = B.T.reshape(r0, N[k], r1)
        cores[0] = self.fibers(0, self.I[0], self.J[0]).reshape(1, N[0], self.r[1])
        return cores, changed

    def interp(self, maxiter=5):
        cores = None
        for it in range(maxiter):
            c1 = self.sweep_lr()
            cores, c2 = self.sweep_rl()
            if not c1 and not c2:
                break
        return cores, it + 1


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    r = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    variant = sys.argv[4] if len(sys.argv) > 4 else "interp"
    extra = int(sys.argv[5]) if len(sys.argv) > 5 else (r if variant != "interp" else 0)
    swap_tol = float(os.environ.get("SWAP_TOL", "0.05"))
    Vs = dense_vstar(n) if n <= 11 else None
    vnorm, vmax = (np.linalg.norm(Vs), np.abs(Vs).max()) if Vs is not None else (1.0, 1.0)
    w = wl.c4_car7d().scaled(ngrid=(n,) * 7, rank=4)
    rcross = int(os.environ.get("CROSS_RANK", str(min(r, n))))
    cr = Cross(w, rcross, variant, extra if variant != "interp" else 0, swap_tol, fit=variant)
    cores = [np.zeros((1 if k == 0 else min(r, n), n, 1 if k == 6 else min(r, n))) for k in range(7)]
    t0 = time.time()
    errs, steps = [], []
    freeze_at = int(os.environ.get("FREEZE", "1000000"))
    for s in range(sweeps):
        if s == freeze_at:
            cr.frozen = True
        W = int(os.environ.get("ANNEAL", "0"))
        if W and s >= 2 * W and s % W == 0 and not cr.frozen:
            a, b = np.median(steps[-W:]), np.median(steps[-2 * W:-W])
            if a > 0.7 * b:
                cr.tol = 1.0 + 2.0 * (cr.tol - 1.0)
                print(f"  anneal at sweep {s}: swap tol -> {cr.tol - 1.0:g}")
        if os.environ.get("TAU_SCHED"):
            tmax, s0 = float(os.environ.get("TAU_MAX", "1.0")), float(os.environ.get("TAU_SCHED"))
            cr.tol = 1.0 + min(tmax, 0.05 * 2.0 ** np.floor(np.log2(1.0 + s / s0)))
        Wn = int(os.environ.get("NOISE_W", "0"))
        if Wn:
            if s == 0:
                snap, path = None, 0.0
            if s > 0:
                path += steps[-1] * np.sqrt(tt_dot(cores, cores))
            if s % Wn == 0:
                if snap is not None:
                    aa_, bb_, ab_ = tt_dot(cores, cores), tt_dot(snap, snap), tt_dot(cores, snap)
                    disp = np.sqrt(max(aa_ - 2 * ab_ + bb_, 0.0))
                    ratio = disp / max(path, 1e-300)
                    if ratio < float(os.environ.get("NOISE_RATIO", "0.5")) and cr.tol - 1.0 < float(os.environ.get("TAU_MAX", "1.0")):
                        cr.tol = 1.0 + 2.0 * (cr.tol - 1.0)
                    print(f"  gauge at sweep {s}: displacement/path {ratio:.3f} -> swap tol {cr.tol - 1.0:g}")
                snap, path = [c.copy() for c in cores], 0.0
        cap = int(os.environ.get("HOLD_CAP", "0"))
        if cap:
            if s == 0:
                hold_left, hold_len = 0, 1
            if hold_left > 0:
                cr.frozen, hold_left = True, hold_left - 1
            else:
                cr.frozen = False
                hold_len = min(2 * hold_len, cap)
                hold_left = hold_len - 1
        cr.set_value(cores)
        new, iters = cr.interp(1 if cr.frozen else 5)
        if rcross > r:
            new = tt_round_rank(new, r)
        aa, bb, ab = tt_dot(new, new), tt_dot(cores, cores), tt_dot(new, cores)
        step = np.sqrt(max(aa - 2 * ab + bb, 0.0)) / np.sqrt(aa)
        steps.append(step)
        cores = new
        if s % 20 == 0 or s == sweeps - 1:
            D = tt_dense(cores) if Vs is not None else None
            e2, em = (np.linalg.norm(D - Vs) / vnorm, np.abs(D - Vs).max() / vmax) if Vs is not None else (np.sqrt(aa), 0.0)
            errs.append(e2)
            print(f"sweep {s:4d} iters {iters} step {step:.3e} err L2 {e2:.3e} max {em:.3e} fibers {cr.nfib} swaps {cr.nswaps} {time.time() - t0:.0f}s", flush=True)
    h = len(steps) // 2
    print(f"variant {variant} extra {cr.p} rank {r}: median step 2nd half {np.median(steps[h:]):.3e}, median err 2nd half {np.median(errs[len(errs) // 2:]):.3e}")


if __name__ == "__main__":
    main()
