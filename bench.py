#!/usr/bin/env python3
"""bench.py — whole-job throughput of the WLSQM hot path on N MI355X of one node.

A "step" is one pass of the fused assemble+factor+solve kernel over one batch of synthetic local
fits (inputs resident in HBM before the timed region starts).  Default workload = BASELINE.json
configs[1] ("C2"): 2D order-2, 1M Halton points, 32 nearest neighbours, WEIGHT_CENTER, all DOFs
unknown.  One process per GPU; for N > 1 the driver launches this file under torch.distributed.run
and every rank fits its own 1M-case shard (independent local problems: no data-path collective,
"weak" scaling).

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))

import synth  # noqa: E402

# BASELINE.json configs -> (dim, order, nk, weighting, knowns); B_fit from SURVEY.md §8d
CONFIGS = {
    "C1": dict(dim=1, order=2, nk=8, wm=1, knowns=0, desc="1D order-2, 8 neighbours, WEIGHT_UNIFORM"),
    "C2": dict(dim=2, order=2, nk=32, wm=2, knowns=0, desc="2D order-2, Halton, 32 neighbours, WEIGHT_CENTER, all DOFs unknown"),
    "C3": dict(dim=2, order=4, nk=64, wm=2, knowns=1, desc="2D order-4, Halton, 64 neighbours, WEIGHT_CENTER, F known"),
    "C5": dict(dim=3, order=2, nk=40, wm=2, knowns=0, desc="3D order-2, Halton, 40 neighbours, WEIGHT_CENTER, all DOFs unknown"),
    # BASELINE configs[3]: the C2 geometry prepared once in an ExpertSolver, then many right-hand sides (run_c4 below)
    "C4": dict(dim=2, order=2, nk=32, wm=2, knowns=0, desc="ExpertSolver, 2D order-2, Halton, 32 neighbours: prepare once + stacked right-hand sides"),
}
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def bytes_per_fit(dim, order, nk, knowns):
    """Algorithmic HBM bytes per fit, dense reference layout:
    8 nk (dim+1) [xk+fk] + 8 dim [xi] + 8 no [fi out] + 8 popcount(knowns) [fi in] + 20 [nk, order, knowns, wm]."""
    no = NDOF[dim][order]
    return 8 * nk * (dim + 1) + 8 * dim + 8 * no + 8 * bin(knowns).count("1") + 20


def load_traffic(config, units_per_launch):
    """HBM bytes per launch from the committed PMC summary (profiles/traffic_<config>.json, tools/make_traffic.py), or None
    when there is none for this config / launch size."""
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % config)
    try:
        t = json.load(open(tfile))
        return t.get("hbm_bytes_per_launch") if int(t.get("cases_per_launch", -1)) == int(units_per_launch) else None
    except Exception:
        return None


def build_problem(cfg, ncases, rank, device=None):
    """Synthetic inputs of SURVEY.md section 8d: Halton points, nk nearest neighbours (self excluded), smooth field.
    With `device` (a torch device) the neighbour search runs on that GPU (wlsqm.hip.knn, exact, same neighbours as
    cKDTree up to the order inside last-ulp distance ties); otherwise scipy's cKDTree on the host."""
    dim, nk = cfg["dim"], cfg["nk"]
    if dim == 1:
        p = synth.line_problem_1d(ncases, nk // 2, seed=rank)
        return p["S"], p["F"], p["hoods"]
    S = synth.halton(ncases, dim, skip=1 + rank * ncases)       # each rank owns a different stretch of the sequence
    F = synth.field(S)
    if device is not None:
        import torch
        import wlsqm.hip as whip
        hoods = whip.knn(torch.from_numpy(S).to(device), nk).cpu().numpy()
        return S, F, hoods
    world = int(os.environ.get("WORLD_SIZE", "1"))
    hoods = synth.knn(S, nk, workers=max(1, len(os.sched_getaffinity(0)) // world))
    return S, F, hoods


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--ncases", type=int, default=1_000_000, help="local fits per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the spot parity check (profiling passes)")
    ap.add_argument("--nrhs", type=int, default=64, help="C4: right-hand sides stacked per step (256 = 4 steps)")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ:        # launched by torch.distributed.run: one rank per GPU, RCCL ("nccl") process group
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import wlsqm.hip as whip
    cfg = CONFIGS[a.config]
    dim, order, nk, n = cfg["dim"], cfg["order"], cfg["nk"], a.ncases
    no = NDOF[dim][order]
    S, F, hoods = build_problem(cfg, n, rank, device=dev)
    if a.config == "C4":
        return run_c4(a, cfg, S, F, hoods, dev, dist, rank, world)

    # device-resident inputs in the reference's dense layout: xk = S[hoods], fk = F[hoods]  (gathered on the GPU)
    S_d = torch.from_numpy(np.ascontiguousarray(S)).to(dev)
    F_d = torch.from_numpy(F).to(dev)
    h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
    xk_d = S_d[h_d].contiguous()
    fk_d = F_d[h_d].contiguous()
    xi_d = S_d.clone()
    fi_d = torch.zeros((n, no), dtype=torch.float64, device=dev)
    fi_d[:, 0] = F_d
    nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev)
    kn_d = torch.full((n,), cfg["knowns"], dtype=torch.int64, device=dev)
    wm_d = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
    del h_d
    args = (dim, order, xk_d, fk_d, nk_d, xi_d, fi_d, kn_d, wm_d)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # The GPU sat idle through the host-side neighbour search above and has dropped to its low power state; the
    # first ~50 ms of kernels run at reduced clocks (measured: 0.214 ms vs 0.183 ms per step).  Bring it back to the
    # clocks of a long-running job before the untimed warm-up, so a short --warmup does not measure the ramp.
    t_wake = time.perf_counter()
    while time.perf_counter() - t_wake < 0.3:
        for _ in range(50):
            whip.fit_many_device(*args)
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        whip.fit_many_device(*args)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        whip.fit_many_device(*args)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel, timed live with HIP events on the stream it is launched on
    ms_kernel = whip.time_fit_device(*args, reps=min(max(a.steps, 20), 500))
    B_fit = bytes_per_fit(dim, order, nk, cfg["knowns"])
    achieved = B_fit * n / (ms_kernel * 1e-3) / 1e9

    out = None
    if rank == 0:
        traffic = load_traffic(a.config, n)
        out = {
            "metric": "local fits/s (whole node)", "value": world * n * a.steps / dt, "unit": "fits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s; %d local fits per GPU per step, device-resident dense xk/fk"
                       % (a.config, cfg["desc"], n), "fits_per_gpu": n, "bytes_per_fit": B_fit},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel_ms": ms_kernel},
        }
        # spot parity check of this very run against the CPU oracle (checker only), judged like the tests:
        # per-column metric, widened only by the oracle's own fp64 noise floor vs an 80-bit solve (tests/_parity.py)
        if a.no_parity:
            print(json.dumps(out), flush=True)
            if dist is not None:
                dist.barrier(); dist.destroy_process_group()
            return
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import oracle
        import _parity
        ns = min(n, 1024)
        fi_g = fi_d[:ns].cpu().numpy()
        xk_h = xk_d[:ns].cpu().numpy(); fk_h = fk_d[:ns].cpu().numpy(); xi_h = xi_d[:ns].cpu().numpy()
        fi_o = np.zeros((ns, no)); fi_o[:, 0] = F[:ns]
        fi_in = fi_o.copy()
        meta = (np.full(ns, nk, np.int32), np.full(ns, order, np.int32), np.full(ns, cfg["knowns"], np.int64),
                np.full(ns, cfg["wm"], np.int32))
        oracle.fit_many(dim, xk_h, fk_h, meta[0], xi_h, fi_o, None, 0, meta[1], meta[2], meta[3], ntasks=8)
        truth = _parity.truth_fit(dim, xk_h, fk_h, meta[0], xi_h, fi_in, meta[1], meta[2], meta[3])
        E = _parity.column_metric(fi_g, fi_o); N = _parity.column_metric(fi_o, truth)
        out["parity"] = {"cases": ns, "colmax_vs_oracle": float(E.max()), "oracle_fp64_noise_floor": float(N.max()),
                         "within_1e-10_plus_8x_noise": bool(np.all(E <= 1e-10 + 8.0 * N))}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(oracle, cfg, xk_d, fk_d, xi_d, F, n, no)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_c4(a, cfg, S, F, hoods, dev, dist, rank, world):
    """BASELINE configs[3]: ExpertSolver on the C2 geometry, prepare once, then right-hand sides F_t = sin(pi x + 0.01 t)
    cos(pi y).  A step = ONE solve_many_device call over --nrhs stacked fields (256 right-hand
    sides = 4 steps at the default 64); a fit = one (case, field) pair.  Algorithmic bytes per fit: fk 8 nk + fi 8 no
    + the geometry (8 nk dim + 8 dim + 20) shared by the nrhs fields of a step.  The time-stepping rate (one fused
    solve_device launch per field, 852 B per fit) is reported beside it."""
    import torch
    import wlsqm
    dim, order, nk, n, R = cfg["dim"], cfg["order"], cfg["nk"], a.ncases, a.nrhs
    no = NDOF[dim][order]
    solver = wlsqm.ExpertSolver(dimension=dim, nk=np.full(n, nk, np.int32), order=np.full(n, order, np.int32),
                                knowns=np.full(n, cfg["knowns"], np.int64),
                                weighting_method=np.full(n, cfg["wm"], np.int32))
    t0 = time.perf_counter()
    solver.prepare(xi=S, xk=S[hoods])
    t_prepare = time.perf_counter() - t0
    S_d = torch.from_numpy(S).to(dev)
    h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
    fk = torch.empty((R, n, nk), dtype=torch.float64, device=dev)
    for r in range(R):
        fk[r] = (torch.sin(np.pi * S_d[:, 0] + 0.01 * r) * torch.cos(np.pi * S_d[:, 1]))[h_d]
    fi = torch.zeros((R, n, no), dtype=torch.float64, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    t_wake = time.perf_counter()
    while time.perf_counter() - t_wake < 0.3:
        solver.solve_many_device(fk, fi)
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        solver.solve_many_device(fk, fi)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        solver.solve_many_device(fk, fi)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the kernel alone, events on the stream it is launched on (torch's current stream is passed to the launch)
    reps = min(max(a.steps, 5), 50)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        solver.solve_many_device(fk, fi)
    e1.record(); torch.cuda.synchronize()
    ms_kernel = e0.elapsed_time(e1) / reps
    # time stepping: one fused launch per field
    fi_seq = torch.zeros_like(fi[0])
    e0.record()
    for r in range(R):
        solver.solve_device(fk[r], fi_seq)
    e1.record(); torch.cuda.synchronize()
    ms_step_field = e0.elapsed_time(e1) / R
    B_fit = 8 * nk + 8 * no + (8 * nk * dim + 8 * dim + 20) / R
    achieved = B_fit * n * R / (ms_kernel * 1e-3) / 1e9
    if rank == 0:
        out = {
            "metric": "local fits/s (whole node)", "value": world * n * R * a.steps / dt, "unit": "fits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4: %s; %d cases x %d stacked fields per GPU per step (a fit = one case of one field), "
                                   "geometry and fields device-resident" % (cfg["desc"], n, R),
                       "fits_per_gpu": n * R, "bytes_per_fit": B_fit, "prepare_ms_host_arrays": t_prepare * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic("C4", n * R), "kernel_ms": ms_kernel},
            "time_stepping": {"ms_per_field": ms_step_field, "fits_per_s": n / (ms_step_field * 1e-3),
                              "bytes_per_fit": bytes_per_fit(dim, order, nk, cfg["knowns"])},
        }
        if a.no_parity:
            print(json.dumps(out), flush=True)
            if dist is not None:
                dist.barrier(); dist.destroy_process_group()
            return
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import oracle
        import _parity
        ns = min(n, 1024)
        r = R - 1
        xk_h = S[hoods[:ns]]; fk_h = fk[r, :ns].cpu().numpy(); xi_h = S[:ns]
        fi_o = np.zeros((ns, no)); fi_in = fi_o.copy()
        meta = (np.full(ns, nk, np.int32), np.full(ns, order, np.int32), np.full(ns, cfg["knowns"], np.int64),
                np.full(ns, cfg["wm"], np.int32))
        oracle.fit_many(dim, xk_h, fk_h, meta[0], xi_h, fi_o, None, 0, meta[1], meta[2], meta[3], ntasks=8)
        truth = _parity.truth_fit(dim, xk_h, fk_h, meta[0], xi_h, fi_in, meta[1], meta[2], meta[3])
        E = _parity.column_metric(fi[r, :ns].cpu().numpy(), fi_o); N = _parity.column_metric(fi_o, truth)
        out["parity"] = {"cases": ns, "field": r, "colmax_vs_oracle": float(E.max()), "oracle_fp64_noise_floor": float(N.max()),
                         "within_1e-10_plus_8x_noise": bool(np.all(E <= 1e-10 + 8.0 * N))}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(oracle, cfg, xk_d, fk_d, xi_d, F, n, no):
    """The CPU oracle (a restatement of the reference's Cython/OpenMP/LAPACK path, kind "port") timed on this
    box's host cores on a bounded sample of the same workload."""
    dim, order, nk = cfg["dim"], cfg["order"], cfg["nk"]
    cores = len(os.sched_getaffinity(0))
    ns = min(n, 400_000)
    xk_h = xk_d[:ns].cpu().numpy(); fk_h = fk_d[:ns].cpu().numpy(); xi_h = xi_d[:ns].cpu().numpy()
    meta = (np.full(ns, nk, np.int32), np.full(ns, order, np.int32), np.full(ns, cfg["knowns"], np.int64),
            np.full(ns, cfg["wm"], np.int32))
    def rate(threads, budget_s):
        best, reps, t_all = 0.0, 0, time.perf_counter()
        while True:
            fi = np.zeros((ns, no)); fi[:, 0] = F[:ns]
            t0 = time.perf_counter()
            oracle.fit_many(dim, xk_h, fk_h, meta[0], xi_h, fi, None, 0, meta[1], meta[2], meta[3], ntasks=threads)
            dt = time.perf_counter() - t0
            best = max(best, ns / dt); reps += 1
            if (reps >= 3 and time.perf_counter() - t_all > budget_s) or reps >= 20:
                return best, reps
    # The restated algorithm keeps the reference's four malloc/free per case (lapackdrivers.pyx:557-560), which stops scaling
    # long before a 256-thread host is full: time a ladder of team sizes and report the best one as the baseline.
    ladder = sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores})
    rates = {}
    for t in ladder:
        rates[t], _ = rate(t, 2.5)
    best_t = max(rates, key=rates.get)
    return {"value": rates[best_t], "unit": "fits/s", "cores": best_t, "kind": "port",
            "sample": "%d cases of the same workload, best pass per team size, OpenMP static schedule; team sizes tried: %s "
                      "(host has %d hardware threads)" % (ns, ", ".join("%d: %.3g" % (t, rates[t]) for t in ladder), cores),
            "value_8_threads": rates.get(8, rates[ladder[0]]), "value_all_threads": rates[cores]}


if __name__ == "__main__":
    main()
