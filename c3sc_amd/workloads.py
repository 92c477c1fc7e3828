"""Workload definitions for the Bellman-backup hot path.

Pure data + seeded generators (numpy only).  Used by bench.py and by the parity tests so the
HIP path and the oracle see bit-identical inputs.  Nothing here touches the GPU or the oracle.

Reference anchors for each problem definition (relative to /root/reference):
  C1 lqg2d   examples/lqg2d_new/lqg2d.c:72-153, 244-245 (bounds +-2), :193 (sigma 1,1)
  C2 dubins  examples/dubinscar_new/dubinscar.c:40-121, 283-322
  C3 lqg6d   examples/lqgnd/lqgnd.c:80-198, 244, 308-309
  C4 car7d   synthetic (the reference has no 7-D car) -- SURVEY.md 8d
  C5 quad10d examples/double_int/double_int.c:80-157, 207, 267-268 with sum-x^2 stage cost
  scar4d     examples/skidding_car/scar.c:40-169, 266-357
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

# model ids -- must match include/c3sc_hip.h (C3SC_MODEL_*)
MODEL_DUBINS3D = 1
MODEL_SCAR4D = 2
MODEL_CAR7D = 3
MODEL_LQGND = 4
MODEL_CHAIN = 5
MODEL_ROSSLER3D = 6
MODEL_TPROB3D = 7
MODEL_PERCH7D = 8
MODEL_SKID5D = 9
MODEL_COTHRUST6D = 10

BC_ABSORB, BC_PERIODIC, BC_REFLECT = 1, 2, 3  # enum EBTYPE, src/boundary.h:42-47
_BC_NAME = {BC_ABSORB: "absorb", BC_PERIODIC: "periodic", BC_REFLECT: "reflect"}


@dataclass
class Workload:
    name: str
    model: int
    params: Tuple[float, ...]
    dx: int
    du: int
    lb: Tuple[float, ...]
    ub: Tuple[float, ...]
    ngrid: Tuple[int, ...]
    ranks: Tuple[int, ...]  # d+1 entries, ranks[0] == ranks[d] == 1
    discount: float
    bc: Tuple[int, ...]
    obstacles: List[Tuple[Tuple[float, ...], Tuple[float, ...]]] = field(default_factory=list)  # (center, widths)
    cands: np.ndarray = None  # (U, du) brute-force control candidates, scan order

    @property
    def dw(self) -> int:
        return self.dx

    @property
    def ncand(self) -> int:
        return int(self.cands.shape[0])

    def bc_names(self):
        return [_BC_NAME[b] for b in self.bc]

    def xgrid(self):
        """linspace(lb, ub, N) per dim, evaluated exactly as C3's linspace (bellman.c:1977)."""
        out = []
        for lo, hi, n in zip(self.lb, self.ub, self.ngrid):
            i = np.arange(n, dtype=np.float64)
            out.append(lo + (hi - lo) * i / float(n - 1))
        return out

    def scaled(self, ngrid=None, rank=None, name=None) -> "Workload":
        """Same problem on a different grid / FT rank (for small parity cases)."""
        ng = tuple(ngrid) if ngrid is not None else self.ngrid
        rk = self.ranks if rank is None else uniform_ranks(self.dx, rank)
        return Workload(name or self.name, self.model, self.params, self.dx, self.du, self.lb, self.ub, ng, rk,
                        self.discount, self.bc, list(self.obstacles), self.cands)


def uniform_ranks(d: int, r: int) -> Tuple[int, ...]:
    return (1,) + (r,) * (d - 1) + (1,)


def _grid_cands(axes) -> np.ndarray:
    return np.array(list(itertools.product(*axes)), dtype=np.float64)


def c1_lqg2d(n=51, r=4) -> Workload:
    return Workload("lqg2d", MODEL_LQGND, (2.0, 1.0, 1.0), 2, 1, (-2.0, -2.0), (2.0, 2.0), (n, n),
                    uniform_ranks(2, r), 0.1, (BC_REFLECT, BC_REFLECT), [],
                    np.linspace(-1.0, 1.0, 33).reshape(-1, 1))


def c2_dubins(n=101, r=6) -> Workload:
    w = 0.5
    return Workload("dubins3d", MODEL_DUBINS3D, (), 3, 1, (-4.0, -4.0, -math.pi), (4.0, 4.0, math.pi), (n, n, n),
                    uniform_ranks(3, r), 0.0, (BC_ABSORB, BC_ABSORB, BC_PERIODIC),
                    [((0.0, 0.0, 0.0), (w, w, 2.0 * math.pi))], np.array([[-1.0], [0.0], [1.0]]))


def c3_lqg6d(n=31, r=8) -> Workload:
    ax = [-1.0, 0.0, 1.0]
    return Workload("lqg6d", MODEL_LQGND, (6.0, 1.0, 1.0), 6, 3, (-2.0,) * 6, (2.0,) * 6, (n,) * 6,
                    uniform_ranks(6, r), 0.1, (BC_REFLECT,) * 6, [], _grid_cands([ax, ax, ax]))


def c4_car7d(n=41, r=10) -> Workload:
    lb = (-4.0, -4.0, -math.pi, 2.0, -2.0, -0.3, -1.0)
    ub = (4.0, 4.0, math.pi, 5.0, 2.0, 0.3, 1.0)
    goal = ((0.0, 0.0, 0.0, 3.5, 0.0, 0.0, 0.0), (1.0, 1.0, 2.0 * math.pi, 3.0, 4.0, 0.6, 2.0))
    bc = (BC_ABSORB, BC_ABSORB, BC_PERIODIC, BC_REFLECT, BC_REFLECT, BC_REFLECT, BC_REFLECT)
    return Workload("car7d", MODEL_CAR7D, (), 7, 2, lb, ub, (n,) * 7, uniform_ranks(7, r), 0.0, bc, [goal],
                    _grid_cands([[-0.5, 0.0, 0.5], [-1.0, 0.0, 1.0]]))


def c5_quad10d(n=25, r=15) -> Workload:
    return Workload("quad10d", MODEL_CHAIN, (10.0, 1.0, 1.0, 1.0), 10, 1, (-2.0,) * 10, (2.0,) * 10, (n,) * 10,
                    uniform_ranks(10, r), 0.1, (BC_REFLECT,) * 10, [], np.array([[-1.0], [0.0], [1.0]]))


def scar4d(n=40, r=20) -> Workload:
    lb = (-4.0, -4.0, -math.pi, 2.0)
    ub = (4.0, 4.0, math.pi, 5.0)
    goal = ((0.0, 0.0, 0.0, 3.5), (1.0, 1.0, 2.0 * math.pi, 3.0))
    ox = np.linspace(-15.0 * math.pi / 180.0, 15.0 * math.pi / 180.0, 3)
    oy = np.linspace(-1.0, 1.0, 3)
    return Workload("scar4d", MODEL_SCAR4D, (), 4, 2, lb, ub, (n,) * 4, uniform_ranks(4, r), 0.0,
                    (BC_ABSORB, BC_ABSORB, BC_PERIODIC, BC_REFLECT), [goal], _grid_cands([ox, oy]))


def rossler3d(n=20, r=8) -> Workload:
    """examples/rossler/rossler.c:208-307: Roessler system with a control on the second equation, [-1,1]^3 with N = 20,
    reflecting box, beta = 0.1, sigma = (1, 1, 1) (its -r / -f options); its BFGS box u in [-4, 4] as a 33-point list
    (set_control_box gives the continuous minimiser over the same box)."""
    return Workload("rossler3d", MODEL_ROSSLER3D, (3.0, 1.0, 1.0), 3, 1, (-1.0,) * 3, (1.0,) * 3, (n,) * 3, uniform_ranks(3, r), 0.1,
                    (BC_REFLECT,) * 3, [], np.linspace(-4.0, 4.0, 33).reshape(-1, 1))


def perch7d(n=20, r=15) -> Workload:
    """examples/perching/perch.c:322-384: glider perching, 7 states, N = 20 per dimension, every face absorbing (the default of
    c3control_create), beta = 1, maxrank 15, one obstacle = the perch (cost 0); its BFGS box u in [-2 pi, 2 pi] as a 41-point
    candidate list, slightly off-centre (set_control_box gives the continuous minimiser over the same box)."""
    lb = (-4.0, -1.0, -math.pi / 2.0, -2.0 * math.pi / 9.0, 0.0, -5.0, -10.0)
    ub = (0.0, 1.0, math.pi / 2.0, 2.0 * math.pi / 9.0, 7.0, 5.0, 10.0)
    goal = ((0.0, 0.0, 0.0, 0.0, 0.0, -2.0, 0.0), (0.1, 0.1, ub[2] - lb[2], ub[3] - lb[3], 0.5, 0.5, ub[6] - lb[6]))
    cands = (np.linspace(-2.0 * math.pi, 2.0 * math.pi, 41) + 0.01).clip(-2.0 * math.pi, 2.0 * math.pi).reshape(-1, 1)
    return Workload("perch7d", MODEL_PERCH7D, (), 7, 1, lb, ub, (n,) * 7, uniform_ranks(7, r), 1.0, (BC_ABSORB,) * 7, [goal], cands)


def tprob3d(n=25, r=10) -> Workload:
    """The 3-state / 3-control problem of the reference's own tests (test/transition_prob/tprob_test.c:2448-2540, Test_bellman_pi3d):
    drift f3, diffusion I, stagecost3d, boundcost 100, box [-1,2] x [-2,3] x [-3,1] with N = 25, every face absorbing (the default of
    c3control_create), beta = 0.1, rank 10; its BFGS box u in [-5, 5]^3 as a 5 x 5 x 5 candidate list (slightly off-centre: no exact
    ties between +u and -u)."""
    ax = np.linspace(-5.0, 5.0, 5) + 0.0625
    return Workload("tprob3d", MODEL_TPROB3D, (), 3, 3, (-1.0, -2.0, -3.0), (2.0, 3.0, 1.0), (n,) * 3, uniform_ranks(3, r), 0.1,
                    (BC_ABSORB,) * 3, [], _grid_cands([ax.clip(-5.0, 5.0)] * 3))


def skid5d(n=40, r=15) -> Workload:
    """examples/skidding5d/scar.c:266-357: the skidding car with lateral dynamics -- (x, y, orientation, yaw rate, lateral speed),
    N = 40 per dimension, x / y reflecting, orientation periodic, the last two dimensions absorbing (c3control_create's default: the
    example leaves them unset, :333-334 are commented out), beta = 1, maxrank 15, the goal box |x|, |y| <= 20 as an obstacle of
    cost 0, 20 steering candidates linspace(-5 pi/180, 5 pi/180) (:290-295)."""
    lb = (-500.0, -500.0, -math.pi, -0.5, -10.0)
    ub = (500.0, 500.0, math.pi, 0.5, 10.0)
    goal = ((0.0, 0.0, 0.0, 0.0, 0.0), (40.0, 40.0, 2.0 * math.pi, 1.0, 20.0))
    bc = (BC_REFLECT, BC_REFLECT, BC_PERIODIC, BC_ABSORB, BC_ABSORB)
    i = np.arange(20, dtype=np.float64)
    cands = (-5.0 * math.pi / 180.0 + (5.0 * math.pi / 180.0 - -5.0 * math.pi / 180.0) * i / 19.0).reshape(-1, 1)  # C3 linspace
    return Workload("skid5d", MODEL_SKID5D, (), 5, 1, lb, ub, (n,) * 5, uniform_ranks(5, r), 1.0, bc, [goal], cands)


def cothrust6d(n=20, r=10) -> Workload:
    """examples/cothrust2/copterposethrust.c:277-383: quadcopter position + velocity, N = 20, every face reflecting, beta = 1,
    rank 10 (start = max), obstacle = the target box around (0, 0, 0, 1, 0, 0); its BFGS box (thrust, roll, pitch) in
    [-1.5, 1.5] x [-0.4, 0.4]^2 as a 4 x 4 x 4 candidate list (64 candidates: the longest list the fiber-pair kernel serves), slightly
    off-centre (set_control_box gives the continuous minimiser over the same box)."""
    lb = (-3.5, -3.5, -2.0, -5.0, -5.0, -5.0)
    ub = (0.2, 3.5, 2.0, 5.0, 5.0, 5.0)
    goal = ((0.0, 0.0, 0.0, 1.0, 0.0, 0.0), (0.4, 0.4, 0.4, 0.5, 0.4, 0.4))
    a0 = (np.linspace(-1.5, 1.5, 4) + 0.011).clip(-1.5, 1.5)
    a1 = (np.linspace(-0.4, 0.4, 4) + 0.003).clip(-0.4, 0.4)
    return Workload("cothrust6d", MODEL_COTHRUST6D, (), 6, 3, lb, ub, (n,) * 6, uniform_ranks(6, r), 1.0, (BC_REFLECT,) * 6, [goal],
                    _grid_cands([a0, a1, a1]))


WORKLOADS = {
    "lqg2d": c1_lqg2d, "dubins3d": c2_dubins, "lqg6d": c3_lqg6d, "car7d": c4_car7d, "quad10d": c5_quad10d,
    "scar4d": scar4d, "rossler3d": rossler3d, "tprob3d": tprob3d, "perch7d": perch7d,
    "skid5d": skid5d, "cothrust6d": cothrust6d,
}

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """First n outputs of the splitmix64 stream started at `seed` (vectorised, wrap-around uint64)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GOLD * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int) -> np.ndarray:
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synth_cores(w: Workload, seed: int = 0xC35C) -> List[np.ndarray]:
    """Seeded nodal TT cores, cores[m] shape (N_m, r_m*r_{m+1}) in the reference layout
    cores[m][j, a + b*r_m]  (valuefunc.c:165-189); values 0.3 + 0.1*U(0,1)  (SURVEY.md 8d)."""
    out = []
    for m in range(w.dx):
        n = w.ngrid[m] * w.ranks[m] * w.ranks[m + 1]
        out.append((0.3 + 0.1 * uniform01(seed + m, n)).reshape(w.ngrid[m], w.ranks[m] * w.ranks[m + 1]))
    return out


def smooth_cores(w: Workload, coef=None) -> List[np.ndarray]:
    """Nodal sampling of sum_m a_m x_m^2 as an exact rank-2 TT, zero-padded to w.ranks."""
    xg = w.xgrid()
    d = w.dx
    a = np.ones(d) if coef is None else np.asarray(coef, dtype=np.float64)
    out = []
    for m in range(d):
        r0, r1 = w.ranks[m], w.ranks[m + 1]
        G = np.zeros((w.ngrid[m], r0, r1))
        q = a[m] * xg[m] ** 2
        if m == 0:
            G[:, 0, 0] = q
            if r1 > 1:
                G[:, 0, 1] = 1.0
            else:
                raise ValueError("rank >= 2 needed")
        elif m == d - 1:
            G[:, 0, 0] = 1.0
            G[:, 1, 0] = q
        else:
            G[:, 0, 0] = 1.0
            G[:, 1, 0] = q
            G[:, 1, 1] = 1.0
        # layout a + b*r0  -> transpose (N, r1, r0) then flatten
        out.append(np.ascontiguousarray(G.transpose(0, 2, 1)).reshape(w.ngrid[m], r0 * r1))
    return out


def synth_fibers(w: Workload, k: int, F: int, seed: int = 0xF1BE) -> np.ndarray:
    """F fibers along dim k: int32 (F, dx) fixed indices drawn uniformly from [0, N_m); entry k is 0."""
    z = splitmix64(seed + k, F * w.dx).reshape(F, w.dx)
    idx = (z % np.array(w.ngrid, dtype=np.uint64)).astype(np.int32)
    idx[:, k] = 0
    return idx


def cross_batch_fibers(w: Workload, k: int, seed: int = 0xF1BE) -> np.ndarray:
    """A TT-cross-shaped batch for core step k: r_k left multi-indices x r_{k+1} right multi-indices."""
    rl, rr = w.ranks[k], w.ranks[k + 1]
    left = synth_fibers(w, k, rl, seed)[:, :k]
    right = synth_fibers(w, k, rr, seed + 977)[:, k + 1:]
    out = np.zeros((rl * rr, w.dx), dtype=np.int32)
    f = 0
    for a in range(rl):
        for b in range(rr):
            out[f, :k] = left[a]
            out[f, k + 1:] = right[b]
            f += 1
    return out


def algorithmic_flops_per_node(w: Workload) -> float:
    """W = W_ft + U*W_mc of SURVEY.md 8d, W_ft averaged uniformly over dim_vary."""
    d, r = w.dx, w.ranks
    tot = 0.0
    for k in range(d):
        n = w.ngrid[k]
        per_node = 4 * r[k] * r[k + 1]
        per_node += sum(2 * r[m] * r[m + 1] for m in range(d) if m != k)
        per_node += sum(4 * r[m + 1] for m in range(k)) + sum(4 * r[m] for m in range(k + 1, d))
        per_fiber = sum(6 * r[m] * r[m + 1] for m in range(d) if m != k)
        tot += per_node + per_fiber / n
    w_ft = tot / d
    w_mc = 16 * d + 10
    return w_ft + w.ncand * w_mc


def algorithmic_bytes_per_node(w: Workload, k: int = 0) -> float:
    """Native batch API: 8 B of output per node + 4*d/N bytes of int32 fixed indices."""
    return 8.0 + 4.0 * w.dx / w.ngrid[k]
