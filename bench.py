#!/usr/bin/env python3
"""bench.py -- Bellman-sweep throughput of the HIP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload car7d] [--fibers F] [--scaling strong|weak]

One "step" = one Bellman sweep over a batch of synthetic fibers: for every varying dimension k = 0..d-1, F fibers
(x N_k nodes) go through the batched bellman_vi kernel (c3sc_hip_bellman_fibers); the sweep ends the way a value-iteration
sweep does: every rank updates the slice of the FT cores it owns from its own outputs, the slices are all-gathered (RCCL
over xGMI when N > 1) and the cores re-staged on the device (valuef_precompute_cores equivalent,
c3sc_hip_upload_value_device).  Inputs (cores, grids, fiber indices) are resident in HBM before the timed region.
Fibers are independent units: the batch is sharded over the ranks in contiguous blocks (c3sc_amd.distributed.shard_range)
with no collective in the data path.  --scaling strong (default): the batch of F fibers per dimension is fixed and split
over the N GPUs (north_star's strong-scaling target); --scaling weak: every rank draws its own F fibers.

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) the script starts its N ranks itself, as fresh child
processes, before anything in the parent touches the GPU.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline`, `cpu_baseline` and -- measured after
the timed region through libc3sc.so on rank 0 at N = 1 -- `vi_sweep` (a whole value-iteration sweep of the solver: ms,
node backups, kernel launches) and `vi_iters_to_tol` (the examples' outer loop under a wall-time budget).  The workload is
BASELINE.json's headline config ("7D car rank-10": SURVEY.md 8d C4 = synthetic 7-D car, 41^7 grid, FT rank 10, 9
brute-force controls).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 vector = FP64 matrix; the microarch guide lists no f64 row
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec
REFERENCE_CPU = {"value": 3.26e5, "unit": "nodes/s", "cores": 1,
                 "what": "the reference's own bellman_vi (src/*.c, gcc -O2 -ftree-vectorize, 1 thread; 5.76e4 with 8 OpenMP threads), "
                         "car7d batch, measured by the survey on the build container's Xeon @2.1 GHz (SURVEY.md section 6); "
                         "context only -- the reference cannot travel to the GPU box, and it cannot be re-measured in this repository: its "
                         "sources need c3/*.h, cdyn/*.h and CBLAS, none of which exist here, and stand-in headers are not allowed (DESIGN.md 2)"}


def _cpu_worker(workload, budget_s, wid):
    """One host core's share of the CPU baseline: runs in its own process (no torch, no GPU), prints one JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from c3sc_amd import workloads as wl

    w = wl.WORKLOADS[workload]()
    P = oracle_lib.Problem(w, wl.synth_cores(w))
    nodes, t0, chunk, k = 0, time.perf_counter(), 256, 0
    while time.perf_counter() - t0 < budget_s:
        idx = wl.synth_fibers(w, k % w.dx, chunk, seed=0xBA5E + 7919 * wid + k)
        P.bellman_fibers(k % w.dx, idx, want_absorbed=False)
        nodes += chunk * w.ngrid[k % w.dx]
        k += 1
    print(json.dumps({"nodes": nodes, "seconds": time.perf_counter() - t0, "chunks": k}), flush=True)


def cpu_baseline(w, cores, budget_s=12.0, nproc=None):
    """Oracle (CPU restatement of the reference algorithm) timed on this box's host cores on a bounded sample of the
    same workload: `nproc` single-threaded worker processes, each on its own fibers for `budget_s` seconds (fibers are
    independent, so this is the fiber-parallel CPU path of SURVEY.md 8d).  The oracle is only the checker / baseline
    here -- never the thing measured as the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    if not os.path.exists(os.path.join(ROOT, "oracle", "libc3sc_oracle.so")):
        oracle_lib.build()
    if nproc is None:
        nproc = max(1, min(16, os.cpu_count() or 1))  # the GPU box's CPU share for one GPU
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", w.name, str(budget_s), str(i)],
                              stdout=subprocess.PIPE, env=env) for i in range(nproc)]
    outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
    nodes = sum(o["nodes"] for o in outs)
    wall = max(o["seconds"] for o in outs)
    per_core = float(np.mean([o["nodes"] / o["seconds"] for o in outs]))
    return {"value": nodes / wall, "unit": "nodes/s", "cores": nproc, "kind": "port", "per_core": per_core,
            "sample": f"{nodes} node backups ({sum(o['chunks'] for o in outs)} chunks of 256 random fibers, dims round-robin) in "
                      f"{wall:.1f} s on {nproc} single-threaded worker processes, oracle/c3sc_oracle.c -O2, "
                      f"{os.cpu_count()} host cores visible",
            "reference_context": REFERENCE_CPU}


def spawn_ranks(n):
    """--gpus N without a launcher: N fresh child processes (one per GPU), started before this process imports torch or
    touches HIP; rank 0's JSON line goes straight to our stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # one rank's failure (a HIP error, an assertion) leaves the others waiting in a collective: poll all of them, and once any child
    # has exited non-zero -- or the whole job overruns its limit -- terminate the rest instead of waiting for them
    rc, deadline = 0, time.time() + float(os.environ.get("C3SC_BENCH_RANK_TIMEOUT", "1500"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is not None:
                alive.remove(p)
                rc = rc or r
        if alive and (rc != 0 or time.time() > deadline):
            for p in alive:
                p.terminate()
            for p in alive:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


def solver_measurements(workload, budget_s):
    """After the timed region, rank 0, one GPU: the solver as a user runs it, through libc3sc.so (reference API names).
    vi_sweep: c3control_step_vi on the bench workload at its rank (own cross driver + batched kernels);
    vi_iters_to_tol: the examples' outer loop (pi_solve(10) + one vi_solve step, e.g. dubinscar.c:343-352) until
    |V_vi - V_pi|_L2 < tol or the wall-time budget ends -- reported as measured, converged or not."""
    import ctypes as C

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import facade_lib
    from c3sc_amd import workloads as wl
    from c3sc_amd.engine import load_library

    H = load_library()
    L = facade_lib.lib()
    for n in ("c3control_init_value", "c3control_step_vi", "c3control_vi_solve", "c3control_pi_solve"):
        getattr(L, n).restype = C.c_void_p
    for n in ("valuef_norm", "valuef_norm2diff"):
        getattr(L, n).restype = C.c_double
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.diag_count.restype = C.c_size_t
    w = wl.WORKLOADS[workload]()
    d = w.dx
    ctl = facade_lib.Control(w, consistent_ends=None)  # the library's default (c3control_set_consistent_ends: on)
    rmax = max(w.ranks)

    def aargs(cross, rnd, kick, start, maxrank, crossrank=0, cross_maxiter=5):
        aa = C.c_void_p(L.approx_args_init())
        L.approx_args_set_crossrank(aa, C.c_size_t(crossrank))  # 0: the cross approximation runs at maxrank (the reference's scheme)
        L.approx_args_set_cross_maxiter(aa, C.c_size_t(cross_maxiter))  # 5: the reference's cap on cross iterations (valuefunc.c:632)
        L.approx_args_set_cross_tol(aa, C.c_double(cross))
        L.approx_args_set_round_tol(aa, C.c_double(rnd))
        L.approx_args_set_kickrank(aa, C.c_size_t(kick))
        L.approx_args_set_startrank(aa, C.c_size_t(start))
        L.approx_args_set_maxrank(aa, C.c_size_t(maxrank))
        return aa

    def smooth(n, x, out, a):
        X = np.ctypeslib.as_array(x, shape=(n, d))
        np.ctypeslib.as_array(out, shape=(n,))[:] = 1.0 + 0.1 * (X ** 2).sum(axis=1)
        return 0

    cb = facade_lib.FIBER_FN(smooth)
    aa = aargs(1e-6, 1e-5, 2, 4, rmax)
    v0 = C.c_void_p(L.c3control_init_value(ctl.h, cb, None, aa, 0))
    ne = C.c_size_t(0)

    def sweeps(nsweeps, host_driven, aa=aa):
        # the same sweeps twice: whole cross iterations on the device (default) and driven from the host (round 2's path)
        if host_driven:
            os.environ["C3SC_HOST_CROSS"] = "1"
        else:
            os.environ.pop("C3SC_HOST_CROSS", None)
        vf = C.c_void_p(L.valuef_copy(v0))
        rows = []
        for it in range(nsweeps):
            l0, t0 = H.c3sc_hip_launch_count(), time.perf_counter()
            nxt = C.c_void_p(L.c3control_step_vi(ctl.h, vf, aa, ctl.opt, 0, C.byref(ne)))
            rows.append((time.perf_counter() - t0, ne.value, H.c3sc_hip_launch_count() - l0))
            L.valuef_destroy(vf)
            vf = nxt
        ranks = [int(L.valuef_get_ranks(vf)[i]) for i in range(d + 1)]
        L.valuef_destroy(vf)
        os.environ.pop("C3SC_HOST_CROSS", None)
        return rows[4:], ranks  # the first sweeps grow the ranks to the cap

    allrows, ranks = sweeps(28, False)
    rows, steady = allrows[:8], allrows[8:]  # sweeps 4..11 (round 2's window) and sweeps 12..27 (the regime a solve spends its time in)
    hrows, _ = sweeps(8, True)
    aa2 = aargs(1e-6, 1e-5, 4, 4, rmax, 2 * rmax)  # cross approximation at twice the rank cap, rounded to the cap (what vi_iters_to_tol runs)
    xrows, xranks = sweeps(20, False, aa2)
    xrows = xrows[6:]
    xms, xnb = 1e3 * float(np.mean([r[0] for r in xrows])), float(np.mean([r[1] for r in xrows]))
    L.approx_args_free(aa2)
    # one cross iteration per sweep (approx_args_set_cross_maxiter(1)): inside a value iteration every sweep warm-starts from the previous
    # sweep's index sets, so the sweeps themselves play the role of the cross iterations -- same step floor, fewer core steps
    one = {}
    for tag, xr1 in (("cross_at_the_cap", 0), ("crossrank2x", 2 * rmax)):
        aa3 = aargs(1e-6, 1e-5, 4, 4, rmax, xr1, 1)
        orows, oranks = sweeps(40, False, aa3)
        orows = orows[12:]  # ranks need more sweeps to reach the cap with one iteration each
        oms, onb = 1e3 * float(np.mean([r[0] for r in orows])), float(np.mean([r[1] for r in orows]))
        one[tag] = {"ms_per_sweep": oms, "median_ms_per_sweep": 1e3 * float(np.median([r[0] for r in orows])), "node_backups_per_sweep": onb, "nodes_per_s_through_the_driver": onb / (oms * 1e-3), "ranks": oranks,
                    "kernel_launches_per_sweep": float(np.mean([r[2] for r in orows]))}
        L.approx_args_free(aa3)
    ms = 1e3 * float(np.mean([r[0] for r in rows]))
    sms = 1e3 * float(np.mean([r[0] for r in steady]))
    snb = float(np.mean([r[1] for r in steady]))
    nb = float(np.mean([r[1] for r in rows]))
    hms = 1e3 * float(np.mean([r[0] for r in hrows]))
    vi_sweep = {"ms_per_sweep": ms, "median_ms_per_sweep": 1e3 * float(np.median([r[0] for r in rows])), "node_backups_per_sweep": nb, "nodes_per_s_through_the_driver": nb / (ms * 1e-3),
                "steady_ms_per_sweep": sms, "steady_node_backups_per_sweep": snb, "steady_nodes_per_s_through_the_driver": snb / (sms * 1e-3),
                "fiber_kernel_launches_per_sweep": float(np.mean([r[2] for r in rows])),
                "cross_iterations_per_sweep": float(np.mean([r[2] for r in rows])) / (2.0 * d),
                "host_driven_ms_per_sweep": hms, "host_driven_node_backups_per_sweep": float(np.mean([r[1] for r in hrows])),
                "ranks": ranks,
                "crossrank2x": {"ms_per_sweep": xms, "node_backups_per_sweep": xnb, "nodes_per_s_through_the_driver": xnb / (xms * 1e-3), "ranks": xranks,
                                "what": f"the same sweeps with approx_args_set_crossrank({2 * rmax}): the cross approximation runs at twice the rank cap "
                                        "(4x the fibers per core step) and its result is rounded to the cap by the TT-SVD; sweeps 10..19 of the series"},
                "one_cross_iteration_per_sweep": dict(one, what="approx_args_set_cross_maxiter(1): sweeps 16..43 of the same series with ONE cross iteration "
                                                               "(left-to-right + right-to-left half sweep, no confirming launch) per value-iteration sweep"),
                "what": f"c3control_step_vi through libc3sc.so on {w.name} (rank cap {rmax}), mean of sweeps 4..11 of a solve from a smooth start (round 2's window): whole cross "
                        "iterations device-resident (c3sc_hip_cross_*: index lists, Bellman launches, node memo, pivoted LU + maxvol per core "
                        "step on one stream); steady_* = the following 16 sweeps of the same series (ranks at their cap, one cross iteration + "
                        "a one-launch confirmation per sweep: where a solve of thousands of sweeps spends its time); host_driven_* = the first "
                        "window again with C3SC_HOST_CROSS=1 (same results bit for bit). "
                        "A sweep now needs ~2 cross iterations instead of 5 (warm-started pivots, exact fixed-point stop), so it backs up "
                        "fewer nodes: the time per sweep is the figure to compare across rounds"}
    L.valuef_destroy(v0)
    L.approx_args_free(aa)

    # "VI iterations to tolerance": c3control_vi_solve's own loop (bellman.c:2282-2340: stop when the L2 step between iterates falls
    # below abs_conv_tol), one sweep per call so that the step series is kept; start value 0
    xr = min(48, int(np.ceil(4.8 * rmax)))  # cross approximation far above the rank cap (the device's core steps end at rank 48), rounded to the cap
    aa = aargs(1e-6, 1e-6, 4, 4, rmax, xr, 1)  # ... and ONE cross iteration per sweep: the value-iteration sweeps are the cross iterations
    zero = facade_lib.FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).fill(0.0), 0)[1])
    cost = C.c_void_p(L.c3control_init_value(ctl.h, zero, None, aa, 0))
    diag = C.c_void_p(None)
    vi_budget = 2.0 * budget_s
    tol_rel, t0, nsw, conv, steps, norms, nb_total = 1e-3, time.perf_counter(), 0, False, [], [], 0
    while time.perf_counter() - t0 < vi_budget:
        nxt = C.c_void_p(L.c3control_step_vi(ctl.h, cost, aa, ctl.opt, 0, C.byref(ne)))
        nb_total += ne.value
        steps.append(L.valuef_norm2diff(cost, nxt))
        L.valuef_destroy(cost)
        cost = nxt
        nsw += 1
        norms.append(L.valuef_norm(cost))
        if steps[-1] < tol_rel * norms[-1]:  # c3control_vi_solve's test (bellman.c:2335) with abs_conv_tol = tol_rel |V|
            conv = True
            break
    t_conv = time.perf_counter() - t0
    # what the iteration does after the tolerance was met: 150 more sweeps, the last 60 of which give the step floor beside the count
    # (right after the stop the true value-iteration step is still a good part of the measured one)
    after = []
    for _ in range(150 if conv else 0):
        nxt = C.c_void_p(L.c3control_step_vi(ctl.h, cost, aa, ctl.opt, 0, C.byref(ne)))
        after.append(L.valuef_norm2diff(cost, nxt) / L.valuef_norm(nxt))
        L.valuef_destroy(cost)
        cost = nxt
    norm = norms[-1] if norms else 0.0
    rel = [sv / nv for sv, nv in zip(steps, norms) if nv > 0]
    iters = {"converged": conv, "tol_rel_L2": tol_rel, "sweeps": nsw, "seconds": t_conv, "ms_per_sweep": 1e3 * t_conv / max(nsw, 1),
             "node_backups": nb_total, "nodes_per_s_through_the_driver": nb_total / t_conv if t_conv > 0 else None, "last_step_L2": steps[-1], "norm_L2": norm, "last_step_rel": steps[-1] / norm if norm else None,
             "step_rel_every_25_sweeps": [float(f"{v:.3e}") for v in rel[::25]],
             "step_rel_after_convergence": {"sweeps_1_to_60": {"median": float(np.median(after[:60])), "min": float(np.min(after[:60])), "max": float(np.max(after[:60]))},
                                            "sweeps_91_to_150": {"median": float(np.median(after[90:])), "min": float(np.min(after[90:])), "max": float(np.max(after[90:]))}} if after else None,
             "first_sweep_with_step_rel_below_1e-2": next((i for i, v in enumerate(rel) if v < 1e-2), None),
             "rank_cap": rmax, "cross_rank": xr, "cross_iterations_per_sweep": 1, "wall_budget_s": vi_budget,
             "end_point_rule": "consistent ends (c3control_set_consistent_ends, the C3Control default; C3SC_LITERAL_ENDS=1 restores nodeutil.c:570-612)",
             "what": "pure value iteration (c3control_vi_solve's loop and stopping test, one sweep per call so that the series is kept) through "
                     "libc3sc.so from the start value 0 until |V_i+1 - V_i|_L2 < tol_rel |V_i+1|_L2.  The value function keeps FT rank 10 (the "
                     f"kernels' input); the cross approximation of T(V) runs at ranks up to {xr} and is cut back to 10 by the TT-SVD "
                     "(approx_args_set_crossrank), one cross iteration per sweep (approx_args_set_cross_maxiter).  On this exit-time problem "
                     "(discount 0, contraction ~1 - 2.5e-3 per sweep) the step floor is the per-sweep noise of the re-selected cross: "
                     "3.6e-3 / 2.5e-3 / 1.3e-3 / 0.95e-3 / 0.69e-3 of |V| at cross rank 10 / 20 / 30 / 40 / 48 (DESIGN.md 6.2, "
                     "profiles/r04_vi_to_tol_car7d_rank10_crossrank*.txt)"}
    L.valuef_destroy(cost)
    L.approx_args_free(aa)
    ctl.close()
    if workload == "car7d":  # is the floor approximation error?  dense ground truth on a reduced grid (a few seconds)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import dense_truth

            iters["reduced_grid_truth"] = dense_truth.reduced_grid_truth(9, 9, 400, crossrank=18)
            iters["reduced_grid_truth_cross_at_the_cap"] = dense_truth.reduced_grid_truth(9, 9, 400, crossrank=0)
        except Exception as e:
            iters["reduced_grid_truth"] = {"error": repr(e)}

    # the same loop on BASELINE config C2 (dubins3d 101^3): the examples' control update (pi_solve(10) + one vi_solve step) to a tolerance
    w2 = wl.WORKLOADS["dubins3d"]()
    ctl = facade_lib.Control(w2, consistent_ends=None)
    aa = aargs(1e-5, 1e-5, 5, 5, 16)
    cost = C.c_void_p(L.c3control_init_value(ctl.h, zero, None, aa, 0))
    tol, t0, diff, outer, conv2 = 2e-2, time.perf_counter(), float("nan"), 0, False
    while time.perf_counter() - t0 < budget_s:
        nxt = C.c_void_p(L.c3control_pi_solve(ctl.h, C.c_size_t(10), C.c_double(1e-2), cost, aa, ctl.opt, 0, C.byref(diag)))
        L.valuef_destroy(cost)
        cost = C.c_void_p(L.c3control_vi_solve(ctl.h, C.c_size_t(1), C.c_double(tol), nxt, aa, ctl.opt, 0, C.byref(diag)))
        diff = L.valuef_norm2diff(nxt, cost)
        L.valuef_destroy(nxt)
        outer += 1
        if diff < tol:
            conv2 = True
            break
    norm2 = L.valuef_norm(cost)
    iters["dubins3d_outer_loop"] = {"converged": conv2, "tol_abs_L2": tol, "outer_iterations": outer, "bellman_sweeps": int(L.diag_count(diag)),
                                    "seconds": time.perf_counter() - t0, "last_diff_L2": diff, "norm_L2": norm2,
                                    "last_diff_rel": diff / norm2 if norm2 else None, "rank_cap": 16,
                                    "what": "dubins3d 101^3: pi_solve(10, 1e-2) + vi_solve(1) per control update (dubinscar.c:343-352) from the start "
                                            "value 0 until |V_vi - V_pi|_L2 < tol"}
    L.diag_destroy(C.byref(diag))
    L.valuef_destroy(cost)
    L.approx_args_free(aa)
    ctl.close()
    return vi_sweep, iters


def pmc_summary(kernel, F, workload):
    """HBM traffic and executed FP64 work of the dominant kernel per launch: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE,
    SQ_INSTS_VALU_*_F64 in separate runs; FETCH x2 per the gfx950 correction of MI355X_MICROARCH.md) cannot be collected
    from inside this process, so the newest committed summary of the same command is read (profiles/README.md)."""
    tag = "fiber_quad" if "fiber_quad" in kernel else ("fiber_pair" if "fiber_pair" in kernel else None)
    if tag is None:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_{tag}_pmc.json")), reverse=True):
        try:
            pm = json.load(open(path))
            if int(pm.get("fibers_per_dim", 0)) != F:
                continue
            ks = [v for kname, v in pm["kernels"].items() if tag in kname]
            if not ks:
                continue
            out = {"traffic": float(np.mean([v["fetch_bytes_x2_gfx950_correction"] + v["write_bytes_per_launch"] for v in ks])),
                   "source": "profiles/" + os.path.basename(path)}
            ex = [v["executed_flops_per_node"] for v in ks if "executed_flops_per_node" in v]
            if ex:
                out["executed_flops_per_node"] = float(np.mean(ex))
            return out
        except (OSError, KeyError, ValueError):
            continue
    return None


def kernel_resources(kernel):
    """Registers / spills / scratch of the timed kernel's instantiations (all K), from the compiler's resource-usage remarks
    collected at build time (c3sc_amd/csrc/kernel_resources.json, tools/kernel_resources.py); None if that file is absent."""
    try:
        res = json.load(open(os.path.join(ROOT, "c3sc_amd", "csrc", "kernel_resources.json")))
    except (OSError, ValueError):
        return None
    import re

    m = re.match(r"(k_fiber_\w+)<(.*?),(\d+),", kernel)
    if not m:
        return None
    fam, model, rp = m.group(1), m.group(2), int(m.group(3))
    hits = [v for k, v in res.items() if k.startswith(fam + "<" + model.replace("<", "<").strip() + ", " + str(rp) + ",") and "true>" not in k.split(",")[-1][:6]]
    if not hits:
        hits = [v for k, v in res.items() if k.startswith(fam + "<" + model + ", " + str(rp) + ",")]
    if not hits:
        return None
    return {"instantiations": len(hits), "vgprs_max": max(h.get("vgprs", 0) for h in hits), "agprs_max": max(h.get("agprs", 0) for h in hits),
            "vgpr_spill_min_max": [min(h.get("vgpr_spill", 0) for h in hits), max(h.get("vgpr_spill", 0) for h in hits)],
            "scratch_bytes_per_lane_min_max": [min(h.get("scratch_bytes_per_lane", 0) for h in hits), max(h.get("scratch_bytes_per_lane", 0) for h in hits)],
            "waves_per_simd": sorted(set(h.get("waves_per_simd", 0) for h in hits)),
            "source": "hipcc -Rpass-analysis=kernel-resource-usage at build time"}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":  # child of cpu_baseline: before anything touches torch / the GPU
        _cpu_worker(sys.argv[2], float(sys.argv[3]), int(sys.argv[4]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="car7d")
    ap.add_argument("--fibers", type=int, default=None,
                    help="fibers per varying dimension per step: the whole job's with --scaling strong, each GPU's with weak "
                         "(default: 2^20 for car7d, the roofline batch of SURVEY.md 8d; 2^17 for the other workloads)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--overlap", action="store_true",
                    help="issue the d independent launches of a step through c3sc_hip_bellman_fibers_all (three streams inside the library) "
                         "in the timed region; per-kernel durations then overlap, so `roofline` is taken from the step span")
    ap.add_argument("--no-overlap-probe", action="store_true", help="skip the untimed second pass that times a step the other way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solver", action="store_true", help="skip the vi_sweep / vi_iters_to_tol measurements after the timed region")
    ap.add_argument("--solver-budget", type=float, default=20.0)
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=None, help="worker processes of the CPU baseline (default min(16, cores))")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    from c3sc_amd import workloads as wl
    from c3sc_amd.distributed import pack_cores, padded_len, shard_range
    from c3sc_amd.engine import BellmanEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # C3SC_BENCH_FORCE_DIST=1 takes the RCCL path (process group, all-gather, barrier, max-reduce) with a single rank too:
    # a rehearsal of the multi-GPU code on a one-GPU box
    use_dist = world > 1 or os.environ.get("C3SC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    json_fd = 1
    if use_dist:
        # RCCL prints a version banner to STDOUT when its first communicator comes up: stdout must carry the one JSON line only,
        # so everything else this process (and the libraries under it) prints goes to stderr from here on
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", device_id=dev)  # RCCL
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    w = wl.WORKLOADS[args.workload]()
    cores = wl.synth_cores(w)
    eng = BellmanEngine(local_rank)
    if args.variant:
        eng.set_variant(args.variant)  # before the value is uploaded: the padded rank follows the variant
    eng.configure(w, cores)

    F = args.fibers if args.fibers is not None else ((1 << 20) if args.workload == "car7d" else (1 << 17))
    d = w.dx
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream
    # fiber batches resident in HBM.  strong: one batch of F fibers per dimension for the whole job, rank r owns the
    # contiguous block shard_range(F, world, r); weak: every rank draws its own F fibers (seed offset by rank)
    if args.scaling == "strong":
        lo, hi = shard_range(F, world, rank)
        idx_t = [torch.from_numpy(np.ascontiguousarray(wl.synth_fibers(w, k, F, seed=0xF1BE)[lo:hi])).to(dev) for k in range(d)]
        F_job = F
    else:
        lo, hi = 0, F
        idx_t = [torch.from_numpy(wl.synth_fibers(w, k, F, seed=0xF1BE + 7919 * rank)).to(dev) for k in range(d)]
        F_job = F * world
    F_loc = hi - lo
    out_t = [torch.empty((F_loc, w.ngrid[k]), dtype=torch.float64, device=dev) for k in range(d)]
    # FT cores on the device in the reference layout (what a cross-approximation step produces); each rank owns a
    # 1/world slice of the flattened cores, updates it from its own outputs and all-gathers the rest per sweep
    flat, offs = pack_cores(cores)
    flat_t = torch.from_numpy(np.concatenate([flat, np.zeros(padded_len(len(flat), world) - len(flat))])).to(dev)
    shard0 = flat_t.view(world, -1)[rank].clone()

    def core_views(buf):
        return [buf[offs[m]:offs[m + 1]] for m in range(d)]

    # the sweep's one collective goes through the library's own RCCL communicator (include/c3sc_hip.h: c3sc_hip_comm_*, what a C
    # main() uses): rank 0's 128-byte id is broadcast once over the process group, the all-gather itself is the C call on the
    # bench's stream, in place in `full_t`
    comm, full_t = None, None
    if use_dist:
        import ctypes as C

        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        # every rank first shows that it can open librccl and make an id (rank 0's is the one used): a rank that cannot must not
        # leave the others waiting inside the communicator's collective initialisation
        idbuf = (C.c_char * 128)()
        can = torch.tensor([1 if eng.L.c3sc_hip_comm_unique_id(idbuf) == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(can, op=dist.ReduceOp.MIN)
        rc_comm = 1
        comm = C.c_void_p()
        if int(can.item()) == 1:
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            idbytes = (C.c_char * 128).from_buffer_copy(bytes(idt.cpu().numpy().tobytes()))
            rc_comm = eng.L.c3sc_hip_comm_create(eng.h, C.c_int(world), C.c_int(rank), idbytes, C.byref(comm))
        # every rank must take the same route: agree on whether all communicators came up; if not, the process group's own
        # all-gather (RCCL through torch.distributed) carries the sweep's collective instead
        okt = torch.tensor([1 if rc_comm == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            if rc_comm == 0:
                eng.L.c3sc_hip_comm_destroy(comm)
            print("bench: c3sc_hip_comm_create failed on some rank (" + eng.L.c3sc_hip_last_error(eng.h).decode() + "); using torch.distributed's all-gather",
                  file=sys.stderr, flush=True)
            comm = None
        else:
            eng.L.c3sc_hip_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        full_t = torch.empty(shard0.numel() * world, dtype=torch.float64, device=dev)

    nodes_per_step_job = sum(F_job * w.ngrid[k] for k in range(d))
    nodes_per_step_loc = sum(F_loc * w.ngrid[k] for k in range(d))
    ev = []

    def step(record, overlap=False):
        if overlap:  # the d independent launches as ONE call: the library spreads them over three streams (c3sc_hip_bellman_fibers_all)
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            eng.bellman_fibers_all(list(range(d)), idx_t, out_t, stream_ptr=sp)
            if record:
                e1.record(stream)  # after the join: the span of the d overlapping launches
                ev.append((-1, e0, e1))
        for k in range(d if not overlap else 0):
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            eng.bellman_fibers(k, idx_t[k], out_t[k], stream_ptr=sp)
            if record:
                e1.record(stream)
                ev.append((k, e0, e1))
        # end of sweep: this rank's slice of the cores becomes a function of its own outputs (a stand-in for the core
        # update of the cross approximation: a true data dependency, values moved by ~1e-12 only so that every step does
        # the same work), the slices are exchanged and the cores re-staged for the next sweep
        probe = out_t[d - 1][: min(F_loc, 4096)].mean()
        if comm is not None:
            mine = full_t.view(world, -1)[rank]
            torch.mul(shard0, 1.0 + 1e-12 * torch.tanh(probe), out=mine)
            rc = eng.L.c3sc_hip_comm_allgather(comm, C.c_void_p(mine.data_ptr()), C.c_void_p(full_t.data_ptr()), C.c_size_t(mine.numel()), C.c_void_p(sp))
            if rc != 0:
                raise SystemExit("c3sc_hip_comm_allgather: " + eng.L.c3sc_hip_last_error(eng.h).decode())
            gathered = full_t
        elif use_dist:
            dist.all_gather_into_tensor(full_t, (shard0 * (1.0 + 1e-12 * torch.tanh(probe))).contiguous())
            gathered = full_t
        else:
            gathered = shard0 * (1.0 + 1e-12 * torch.tanh(probe))
        eng.upload_value_device(w.ranks, core_views(gathered), sp)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False, args.overlap)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True, args.overlap)
    fence()
    elapsed = time.perf_counter() - t0
    # the same steps the other way (untimed for `value`): with the launches of a step on one stream (default: the per-kernel
    # durations behind `roofline` are then what rocprofv3 sees) or spread over three streams by the library (--overlap)
    other = None
    if not use_dist and not args.no_overlap_probe:
        step(False, not args.overlap)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(False, not args.overlap)
        fence()
        other = (time.perf_counter() - t1) / args.steps
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    status = eng.status()

    # dominant kernel: average launch duration from the HIP events recorded on the launch stream
    kms = [e0.elapsed_time(e1) / (d if kk < 0 else 1) for (kk, e0, e1) in ev]  # --overlap: span of a step's launches / their number
    avg_ms = float(np.mean(kms))
    # per varying dimension (one kernel instantiation each): what a rocprofv3 kernel_stats row of the same run is compared with
    ms_by_dim = [float(np.mean([e0.elapsed_time(e1) for (kk, e0, e1) in ev if kk == k])) for k in range(d)] if not args.overlap else None
    Wf = wl.algorithmic_flops_per_node(w)
    nodes_per_launch = nodes_per_step_loc / d
    achieved_tflops = Wf * nodes_per_launch / (avg_ms * 1e-3) / 1e12
    bytes_per_node = float(np.mean([wl.algorithmic_bytes_per_node(w, k) for k in range(d)]))
    hbm_gbs = bytes_per_node * nodes_per_launch / (avg_ms * 1e-3) / 1e9
    kern = eng.last_kernel()
    kres = kernel_resources(kern)
    pm = pmc_summary(kern, F_loc, w.name)
    if "K=" in kern:
        kern = kern[: kern.index("K=")] + "K=0..%d>" % (d - 1)

    if rank == 0:
        res = {
            "metric": "Bellman-sweep nodes/sec (7D car rank-10)" if args.workload == "car7d" else f"Bellman-sweep nodes/sec ({w.name})",
            "value": nodes_per_step_job * args.steps / elapsed,
            "unit": "nodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{w.name}: d={w.dx} N={w.ngrid[0]} FT rank {max(w.ranks)} U={w.ncand} controls, "
                                   f"{F_job} random fibers per varying dim per step for the whole job ({nodes_per_step_job} node backups/step), "
                                   f"seeded synthetic cores (SURVEY.md 8d {'C4' if w.name == 'car7d' else ''})",
                       "fibers_per_dim_per_gpu": F_loc, "fibers_per_dim_job": F_job,
                       "parallelism": f"fiber-sharded x{world} ({args.scaling} scaling)" if world > 1 else "single GPU",
                       "exchange": "all-gather of the updated FT cores per sweep (RCCL)" if world > 1 else "none"},
            "kernel_status_flags": status,
            "roofline": {
                "bound": "valu_fp64", "achieved": achieved_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / FP64_PEAK_TFLOPS,
                "traffic": pm["traffic"] if pm else None, "traffic_unit": "bytes/launch", "traffic_source": pm["source"] if pm else None,
                "algorithmic_bytes_per_launch": bytes_per_node * nodes_per_launch,
                "kernel": kern, "kernel_resources": kres, "avg_launch_ms": avg_ms, "launches": len(kms), "launch_ms_by_dim": ms_by_dim,
                "algorithmic_flops_per_node": Wf, "nodes_per_launch": nodes_per_launch,
                # what the kernel actually executes (fold-once algebra), from the SQ_INSTS_VALU_*_F64 / MFMA counters of the
                # committed PMC pass of this command; null when no such pass is committed for this kernel and batch
                "executed_flops_per_node": pm.get("executed_flops_per_node") if pm else None,
                "executed_frac": (pm["executed_flops_per_node"] * nodes_per_launch / (avg_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS)
                if pm and "executed_flops_per_node" in pm else None,
                "traffic_note": "WRITE_SIZE above the output bytes is scratch (spilled registers, kernel_resources), not partial-line stores: DESIGN.md 4.1",
                "note": "`achieved`/`frac` credit the ALGORITHMIC flops of SURVEY.md 8d (W = W_ft + U W_mc per node) as the task "
                        "defines them; the kernel's fold-once algebra executes fewer (executed_flops_per_node, executed_frac). "
                        "FP64 vector and matrix peak are the same 78.6 TFLOP/s datasheet figure",
                "hbm_secondary": {"algorithmic_bytes_per_node": bytes_per_node, "achieved_GBs": hbm_gbs,
                                  "peak_GBs": HBM_PEAK_GBS, "frac": hbm_gbs / HBM_PEAK_GBS},
            },
        }
        # the step the other way round, measured after the timed region (never `value`): the d launches of a step are independent,
        # and c3sc_hip_bellman_fibers_all spreads them over three streams so that one dimension's tail overlaps the next one's head
        res["multi_gpu_notes"] = {
            "launch_mode": "the d launches of a step stay on ONE stream at every N (default): per-kernel durations are then what HIP events and "
                           "rocprofv3 both see, which `roofline` is defined on; --overlap (c3sc_hip_bellman_fibers_all, three streams) must be "
                           "given at every N or at none for a like-for-like scaling curve -- it lifts the 2^17-fibers-per-rank launch of an "
                           "8-GPU strong-scaling run to the per-fiber rate of the 2^20 batch",
            "sweep_end_core_update": "stand-in (data-dependent rescaling of the rank's slice of the cores, then the all-gather and the re-staging): "
                                     "the roofline batch is 2^20 RANDOM fibers per dimension without cross structure, a real core step "
                                     "(k_cross_core) on it would factor a meaningless matrix; the real core steps are timed through the solver "
                                     "(vi_sweep) and run sharded under c3control_shard_over_gpus",
            "failure_handling": "a rank whose launch or staging copy fails still enters the all-gather (rows marked NaN) and all ranks stop "
                                "together (cross_device.hip step_fibers, comm_rccl.hip c3sc_hip_comm_exchange; tests/test_distributed.py)"}
        res["step_launch_mode"] = "overlapped (c3sc_hip_bellman_fibers_all, three streams)" if args.overlap else "one stream, d launches in order"
        if other is not None:
            res["other_mode_step"] = {
                "mode": "one stream, d launches in order" if args.overlap else "overlapped (c3sc_hip_bellman_fibers_all, three streams)",
                "ms_per_step": 1e3 * other, "value": nodes_per_step_job / other, "unit": "nodes/s",
                "roofline_frac_from_step_span": Wf * nodes_per_step_loc / other / 1e12 / FP64_PEAK_TFLOPS,
                "what": "the same steps re-timed (wall clock over --steps steps, untimed for `value`) with the other launch mode; under "
                        "overlap the per-kernel durations of a profile overlap, so the default keeps one stream and `roofline` per kernel"}
        res["vi_sweep"], res["vi_iters_to_tol"] = None, None
        if world == 1 and not args.no_solver:
            try:
                res["vi_sweep"], res["vi_iters_to_tol"] = solver_measurements(args.workload, args.solver_budget)
            except Exception as e:  # the headline measurement above stands on its own
                res["vi_sweep"] = {"error": repr(e)}
        res["cpu_baseline"] = cpu_baseline(w, cores, args.cpu_budget, args.cpu_procs) if (world == 1 and not args.no_cpu_baseline) else None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        if comm is not None:
            eng.L.c3sc_hip_comm_destroy(comm)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
