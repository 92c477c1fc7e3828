#!/usr/bin/env python3
"""bench.py -- Bellman-sweep throughput of the HIP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload car7d] [--fibers F]

One "step" = one Bellman sweep over a batch of synthetic fibers: for every varying dimension
k = 0..d-1, F fibers (x N_k nodes) go through the batched bellman_vi kernel
(c3sc_hip_bellman_fibers), then the sweep ends the way a value-iteration sweep does: the updated FT
cores are exchanged (RCCL all-gather over xGMI when N > 1) and re-staged on the device
(valuef_precompute_cores equivalent, c3sc_hip_upload_value_device).  Inputs (cores, grids, fiber
indices) are resident in HBM before the timed region.  Fibers are independent units: each rank owns
its own F fibers per dimension (weak scaling), no collective in the data path.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and
`cpu_baseline` objects.  The workload is BASELINE.json's headline config ("7D car rank-10":
SURVEY.md 8d C4 = synthetic 7-D car, 41^7 grid, FT rank 10, 9 brute-force controls).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 vector = FP64 matrix; the microarch guide lists no f64 row
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec


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
    import subprocess

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
                      f"{os.cpu_count()} host cores visible"}


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
                    help="fibers per varying dimension per GPU per step (default: 2^20 for car7d, the roofline batch of "
                         "SURVEY.md 8d; 2^17 for the other workloads)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=None, help="worker processes of the CPU baseline (default min(16, cores))")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from c3sc_amd import workloads as wl
    from c3sc_amd.distributed import allgather_cores, pack_cores, padded_len
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
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)  # RCCL
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    w = wl.WORKLOADS[args.workload]()
    cores = wl.synth_cores(w)
    eng = BellmanEngine(local_rank)
    eng.configure(w, cores)
    if args.variant:
        eng.set_variant(args.variant)

    F = args.fibers if args.fibers is not None else ((1 << 20) if args.workload == "car7d" else (1 << 17))
    d = w.dx
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream
    # fiber batches resident in HBM; every rank owns different fibers (seed offset by rank)
    idx_t = [torch.from_numpy(wl.synth_fibers(w, k, F, seed=0xF1BE + 7919 * rank)).to(dev) for k in range(d)]
    out_t = [torch.empty((F, w.ngrid[k]), dtype=torch.float64, device=dev) for k in range(d)]
    # FT cores on the device in the reference layout (what a cross-approximation step produces);
    # each rank "owns" a 1/world slice of the flattened cores and all-gathers the rest per sweep
    flat, offs = pack_cores(cores)
    flat_t = torch.from_numpy(np.concatenate([flat, np.zeros(padded_len(len(flat), world) - len(flat))])).to(dev)
    shard = flat_t.view(world, -1)[rank].clone()

    def core_views(buf):
        return [buf[offs[m]:offs[m + 1]] for m in range(d)]

    nodes_per_step = sum(F * w.ngrid[k] for k in range(d))
    ev = []

    def step(record):
        for k in range(d):
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            eng.bellman_fibers(k, idx_t[k], out_t[k], stream_ptr=sp)
            if record:
                e1.record(stream)
                ev.append((k, e0, e1))
        # end of sweep: exchange the updated cores and re-stage them for the next sweep
        gathered = allgather_cores(shard, world, force=use_dist) if use_dist else flat_t
        eng.upload_value_device(w.ranks, core_views(gathered), sp)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    status = eng.status()

    # dominant kernel: average launch duration from the HIP events recorded on the launch stream
    kms = [e0.elapsed_time(e1) for (_, e0, e1) in ev]
    avg_ms = float(np.mean(kms))
    Wf = wl.algorithmic_flops_per_node(w)
    nodes_per_launch = nodes_per_step / d
    achieved_tflops = Wf * nodes_per_launch / (avg_ms * 1e-3) / 1e12
    bytes_per_node = float(np.mean([wl.algorithmic_bytes_per_node(w, k) for k in range(d)]))
    hbm_gbs = bytes_per_node * nodes_per_launch / (avg_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel per launch: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate
    # runs; FETCH x2 per the gfx950 correction of MI355X_MICROARCH.md) cannot be collected from inside this
    # process, so the committed summary of the same command is read (profiles/README.md says how it was made).
    traffic, traffic_src = None, None
    try:
        pmc_file = {1 << 17: "r01_f_fiber_pair_pmc.json", 1 << 20: "r01_g_fiber_pair_pmc.json"}.get(F)
        if args.workload == "car7d" and pmc_file:
            pm = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            ks = [v for kname, v in pm["kernels"].items() if "k_fiber_pair" in kname]
            if ks and "fiber_pair" in eng.last_kernel():
                traffic = float(np.mean([v["fetch_bytes_x2_gfx950_correction"] + v["write_bytes_per_launch"] for v in ks]))
                traffic_src = "profiles/" + pmc_file + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch, FETCH x2)"
    except (OSError, KeyError, ValueError):
        pass
    kern = eng.last_kernel()
    if "K=" in kern:
        kern = kern[: kern.index("K=")] + "K=0..%d>" % (d - 1)

    if rank == 0:
        res = {
            "metric": "Bellman-sweep nodes/sec (7D car rank-10)" if args.workload == "car7d" else f"Bellman-sweep nodes/sec ({w.name})",
            "value": nodes_per_step * args.steps * world / elapsed,
            "unit": "nodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{w.name}: d={w.dx} N={w.ngrid[0]} FT rank {max(w.ranks)} U={w.ncand} controls, "
                                   f"{F} random fibers per varying dim per GPU per step ({nodes_per_step} node backups/GPU/step), "
                                   f"seeded synthetic cores (SURVEY.md 8d {'C4' if w.name == 'car7d' else ''})",
                       "fibers_per_dim_per_gpu": F, "parallelism": f"fiber-sharded x{world}" if world > 1 else "single GPU",
                       "exchange": "all-gather of FT cores per sweep (RCCL)" if world > 1 else "none"},
            # not measured by this run: the examples' outer loop through libc3sc.so on the same config (tools/solve_to_tol.py)
            "vi_iters_to_tol": ({"outer_iterations": 30, "bellman_sweeps": 341, "seconds": 5.6,
                                 "criterion": "|V| plateau reached; the step difference then stays at the rank-10 truncation floor (0.4-1.7 % of |V|)",
                                 "source": "profiles/r01_f_solve_car7d.txt"} if args.workload == "car7d" else None),
            "kernel_status_flags": status,
            "roofline": {
                "bound": "mfma", "achieved": achieved_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_per_node * nodes_per_launch,
                "kernel": kern, "avg_launch_ms": avg_ms, "launches": len(kms),
                "algorithmic_flops_per_node": Wf, "nodes_per_launch": nodes_per_launch,
                "note": "FP64 compute bound (vector FMA path; dense f64 MFMA peak is the same 78.6 TFLOP/s datasheet figure)",
                "hbm_secondary": {"algorithmic_bytes_per_node": bytes_per_node, "achieved_GBs": hbm_gbs,
                                  "peak_GBs": HBM_PEAK_GBS, "frac": hbm_gbs / HBM_PEAK_GBS},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(w, cores, args.cpu_budget, args.cpu_procs)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
