#!/usr/bin/env python3
"""Does the device-resident API capture into a HIP graph (torch.cuda.CUDAGraph)?  Small batches, where launch overhead shows."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip

dev = torch.device("cuda", 0)
for cfgname, n in (("C2", 4096), ("C2", 65536), ("C3", 4096), ("C5", 16384)):
    cfg = bench.CONFIGS[cfgname]
    dim, order, nk = cfg["dim"], cfg["order"], cfg["nk"]
    no = bench.NDOF[dim][order]
    S, F, hoods = bench.build_problem(cfg, n, 0, device=dev)
    S_d = torch.from_numpy(np.ascontiguousarray(S)).to(dev); F_d = torch.from_numpy(F).to(dev)
    h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
    xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous(); xi = S_d.clone()
    fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
    nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev)
    kn = torch.full((n,), cfg["knowns"], dtype=torch.int64, device=dev)
    wm = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
    args = (dim, order, xk, fk, nk_d, xi, fi, kn, wm)
    STEPS = 50
    whip.fit_many_device(*args); torch.cuda.synchronize()
    ref = fi.clone()
    t0 = time.perf_counter()
    for _ in range(STEPS * 4): whip.fit_many_device(*args)
    torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / (STEPS * 4)
    side = torch.cuda.Stream()
    fi.zero_(); fi[:, 0] = F_d
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            for _ in range(STEPS): whip.fit_many_device(*args)
    except Exception as e:
        print(cfgname, n, "capture failed:", repr(e)[:300]); continue
    torch.cuda.synchronize()
    captured_ran = not torch.equal(fi, torch.zeros_like(fi).index_put_((torch.arange(n, device=dev), torch.zeros(n, dtype=torch.long, device=dev)), F_d))
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(fi, ref)
    t0 = time.perf_counter()
    for _ in range(4): g.replay()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / (STEPS * 4)
    print("%s n=%d: eager %.1f us per call, graph replay %.1f us per call; ran during capture: %s; replay bit-identical: %s"
          % (cfgname, n, t_eager * 1e6, t_graph * 1e6, captured_ran, ok))
