"""Generate tests/golden/car7d_small.npz and tests/golden/pi_dubins_small.npz from the pinned CPU oracle (run once;
output committed).
The fixture is data only: seeded inputs are regenerated from c3sc_amd.workloads, the file holds the
fiber indices and the oracle's outputs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib  # noqa: E402
from c3sc_amd import workloads as wl  # noqa: E402

oracle_lib.build()
ngrid, rank = (9, 8, 10, 7, 6, 5, 11), 4
w = wl.c4_car7d().scaled(ngrid=ngrid, rank=rank)
P = oracle_lib.Problem(w, wl.synth_cores(w))
data = {"ngrid": np.array(ngrid), "rank": np.array(rank)}
for k in range(w.dx):
    idx = wl.synth_fibers(w, k, 16, seed=0x601D)
    idx[0, :] = 0
    idx[1, :] = np.array(ngrid) - 1
    idx[:, k] = 0
    out, ui, ab = P.bellman_fibers(k, idx)
    data[f"idx{k}"], data[f"out{k}"], data[f"ui{k}"], data[f"ab{k}"] = idx, out, ui, ab
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "car7d_small.npz"), **data)
print("wrote car7d_small.npz")


# policy evaluation (bellman_pi, bellman.c:1702-1886): policy greedy for one value function, right-hand side on another
ngrid2, rank2 = (21, 17, 16), 4
w2 = wl.c2_dubins().scaled(ngrid=ngrid2, rank=rank2)
cores_pol = wl.synth_cores(w2)
cores_it = wl.smooth_cores(w2)
P2 = oracle_lib.Problem(w2, cores_it)
pol_vf = oracle_lib.ValueF(w2.ngrid, w2.ranks, cores_pol)
Ppol = oracle_lib.Problem(w2, cores_pol)
data2 = {"ngrid": np.array(ngrid2), "rank": np.array(rank2)}
for k in range(w2.dx):
    idx = wl.synth_fibers(w2, k, 12, seed=0x601E)
    idx[0, :] = 0
    idx[1, :] = np.array(ngrid2) - 1
    idx[:, k] = 0
    _, ui, ab = Ppol.bellman_fibers(k, idx)          # the greedy policy (candidate index per node)
    P2.pi_begin()                                     # fresh tables: nothing cached, every policy is computed here
    P2.pi_step_begin()
    out, ui2 = P2.policy_fibers(pol_vf, k, idx)
    assert ((ui2 == ui) | (ui2 < 0)).all()
    data2[f"idx{k}"], data2[f"policy{k}"], data2[f"out{k}"], data2[f"ab{k}"] = idx, ui, out, ab
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pi_dubins_small.npz"), **data2)
print("wrote pi_dubins_small.npz")


# ---------------------------------------------------------------------------------------------------------------
# tests/golden/path_pieces.npz -- the fixture families SURVEY.md 8c lists for the pieces of the path (G1-G4, G6, G7).
# Inputs and expected outputs of the pinned oracle; tests/test_golden_pieces.py replays them against the oracle, the
# host library and (stencils) the device.  Seeds are fixed; arrays are stored flat with an index of (offset, shape).
def _pieces():
    rng = np.random.default_rng(0x60D5)
    rec = {}
    meta = []

    def put(name, arr):
        rec[name] = np.ascontiguousarray(arr)

    # G1 transition_assemble (no-grad and grad), incl. dead zone, stationary (ret 1) and ambiguous-gradient (ret 2) cases
    n1 = 0
    for dx in (2, 3, 4, 7, 10):
        for case in range(6):
            du = int(rng.integers(1, 4))
            tv = rng.uniform(1e-3, 2.0, 2 * dx)
            drift = rng.uniform(-2, 2, dx)
            if case >= 1:
                drift[rng.integers(0, dx)] = 1e-14 * rng.choice([-1.0, 1.0, 0.5])  # inside / on the dead zone
            diff = np.diag(rng.uniform(0.1, 1.0, dx)).ravel()
            gd = rng.uniform(-1, 1, dx * du)
            gdiff = rng.uniform(-0.1, 0.1, dx * dx * du)
            if case == 4:
                drift[:] = 0.0
                diff[:] = 0.0  # Q < 1e-14: return 1, outputs untouched
            if case == 5:
                drift[0] = 0.0
                gd[0::dx] = 0.0  # zero drift with zero gradient in dim 0: return 2
            h2 = float(rng.uniform(1e-4, 1e-2))
            r0, p0, dt0, _, _ = oracle_lib.transition_assemble(dx, du, dx, h2, tv, drift, diff)
            r1, p1, dt1, gp1, gdt1 = oracle_lib.transition_assemble(dx, du, dx, h2, tv, drift, diff, gd, gdiff)
            for k, v in dict(dx=dx, du=du, h2=h2, tv=tv, drift=drift, diff=diff, gd=gd, gdiff=gdiff, ret=r0, prob=p0, dt=dt0,
                             ret_g=r1, prob_g=p1, dt_g=dt1, gprob=gp1, gdt=gdt1).items():
                put(f"g1_{n1}_{k}", np.asarray(v))
            n1 += 1
    put("g1_n", np.array(n1))

    # G2 bellmanrhs (value and gradient)
    n2 = 0
    for dx in (2, 3, 7):
        for _ in range(4):
            du = int(rng.integers(1, 4))
            S = 2 * dx + 1
            prob = rng.uniform(0, 1, S); prob /= prob.sum()
            cost = rng.uniform(0, 5, S)
            stage, disc, dt = float(rng.uniform(0, 3)), float(rng.choice([0.0, 0.1, 2.0])), float(rng.uniform(1e-4, 1e-1))
            sg, pg, dtg = rng.uniform(-1, 1, du), rng.uniform(-1, 1, S * du), rng.uniform(-1, 1, du)
            v0, _ = oracle_lib.bellmanrhs(dx, du, stage, disc, prob, dt, cost)
            v1, g1 = oracle_lib.bellmanrhs(dx, du, stage, disc, prob, dt, cost, sg, pg, dtg)
            for k, v in dict(dx=dx, du=du, stage=stage, disc=disc, dt=dt, prob=prob, cost=cost, sg=sg, pg=pg, dtg=dtg, val=v0,
                             val_g=v1, grad=g1).items():
                put(f"g2_{n2}_{k}", np.asarray(v))
            n2 += 1
    put("g2_n", np.array(n2))

    # G3 convert_fiber_to_ind + process_fibers_neighbor over boundary-type combinations, obstacle, face fibers
    n3 = 0
    names = {1: "absorb", 2: "periodic", 3: "reflect"}
    for d, ngrid in ((2, (7, 9)), (3, (6, 5, 8)), (4, (5, 4, 6, 5))):
        lb, ub = -np.ones(d), np.linspace(1.0, 2.0, d)
        xg = [lb[m] + (ub[m] - lb[m]) * np.arange(ngrid[m]) / float(ngrid[m] - 1) for m in range(d)]
        for combo in range(9 if d == 2 else (14 if d == 3 else 8)):
            bc = [1 + (combo // 3 ** m) % 3 for m in range(d)] if d == 2 else [int(v) for v in rng.integers(1, 4, d)]
            bnd = oracle_lib.Boundary(lb, ub)
            for m in range(d):
                bnd.set_type(m, names[bc[m]])
            obs = combo % 2 == 1
            if obs:
                bnd.add_obstacle(np.zeros(d) + 0.1, np.full(d, 0.9))
            for k in range(d):
                fi = np.array([int(rng.integers(0, ngrid[m])) for m in range(d)])
                if rng.random() < 0.4:
                    m = int(rng.integers(0, d)); fi[m] = int(rng.choice([0, ngrid[m] - 1]))
                N = ngrid[k]
                x = np.zeros((N, d))
                for m in range(d):
                    x[:, m] = xg[m][fi[m]]
                x[:, k] = xg[k]
                res, fi0, dv = oracle_lib.convert_fiber_to_ind(x, ngrid, xg)
                _, ab, nv, nf = oracle_lib.process_fibers_neighbor(fi0, dv, x, ngrid, bnd)
                for kk, v in dict(d=d, ngrid=ngrid, lb=lb, ub=ub, bc=bc, obs=int(obs), k=k, x=x, ret=res, fi=fi0.astype(np.int64), dv=dv,
                                  ab=ab, nv=nv.astype(np.int64), nf=nf.astype(np.int64)).items():
                    put(f"g3_{n3}_{kk}", np.asarray(v))
                n3 += 1
    put("g3_n", np.array(n3))

    # G4 valuef_eval_fiber_ind_nn: seeded random cores, mixed ranks, every dim_vary
    n4 = 0
    for d, ngrid, ranks in ((2, (6, 7), (1, 3, 1)), (3, (5, 6, 4), (1, 2, 4, 1)), (6, (4, 5, 3, 4, 5, 3), (1, 2, 3, 2, 3, 2, 1))):
        cores = [rng.uniform(-1, 1, (ngrid[m], ranks[m] * ranks[m + 1])) for m in range(d)]
        vf = oracle_lib.ValueF(ngrid, ranks, cores)
        for k in range(d):
            fi = np.array([int(rng.integers(0, ngrid[m])) for m in range(d)]); fi[k] = 0
            nbf = []
            for m in range(d):
                if m != k:
                    nbf += [int(rng.integers(0, ngrid[m])), int(rng.integers(0, ngrid[m]))]
            nbv = rng.integers(0, ngrid[k], 2 * ngrid[k])
            out = vf.eval_fiber_ind_nn(fi, k, np.array(nbf), nbv)
            put(f"g4_{n4}_d", np.array(d)); put(f"g4_{n4}_ngrid", np.array(ngrid)); put(f"g4_{n4}_ranks", np.array(ranks))
            for m in range(d):
                put(f"g4_{n4}_core{m}", cores[m])
            put(f"g4_{n4}_k", np.array(k)); put(f"g4_{n4}_fi", fi); put(f"g4_{n4}_nbf", np.array(nbf)); put(f"g4_{n4}_nbv", nbv)
            put(f"g4_{n4}_out", out)
            n4 += 1
    put("g4_n", np.array(n4))

    # G6 memo keys: decimal strings and buckets, incl. multi-digit indices and iteration counters >= 10
    tuples = [[0, 0, 0, 0], [3, 17, 0, 1], [100, 99, 7, 0, 12], [40, 40, 40, 40, 40, 40, 40, 0, 123], [5, 0, 9, 0, 1000], [65535, 1, 0, 10]]
    put("g6_n", np.array(len(tuples)))
    for i, t in enumerate(tuples):
        s = oracle_lib.key_string(np.array(t))
        put(f"g6_{i}_tuple", np.array(t)); put(f"g6_{i}_key", np.frombuffer(s, dtype=np.uint8))
        put(f"g6_{i}_bucket", np.array(oracle_lib.hashchar(1000000, s)))

    # G7 grid constants of every configuration: linspace grids, h, h_min^2, (t, t2) pairs
    cfgs = ["lqg2d", "dubins3d", "lqg6d", "car7d", "quad10d", "scar4d"]
    put("g7_n", np.array(len(cfgs)))
    for i, name in enumerate(cfgs):
        w = wl.WORKLOADS[name]()
        P = oracle_lib.Problem(w, wl.synth_cores(w))
        put(f"g7_{i}_name", np.frombuffer(name.encode(), dtype=np.uint8))
        for m in range(w.dx):
            put(f"g7_{i}_grid{m}", P.xgrid(m))
        put(f"g7_{i}_h2", np.array(P.h2())); put(f"g7_{i}_t", P.tvec())
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_pieces.npz"), **rec)
    print("wrote path_pieces.npz:", n1, "G1,", n2, "G2,", n3, "G3,", n4, "G4 cases")


_pieces()
