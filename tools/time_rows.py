#!/usr/bin/env python3
"""Time the basic fit of the 3D order-3 / order-4 systems (20 / 35 unknowns: csrc/fit_rows.hip) on device-resident dense input.
usage: python tools/time_rows.py [ncases] [K] [sorted]   ("sorted": neighbours by ascending distance, as a k-nearest-neighbour search delivers them)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import wlsqm.hip as whip
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
by_distance = len(sys.argv) > 3 and sys.argv[3] == "sorted"
for order, no in ((3, 20), (4, 35)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    xi = torch.rand((n, 3), dtype=torch.float64, device=dev, generator=g)
    xk = (xi[:, None, :] + 0.05 * (2 * torch.rand((n, K, 3), dtype=torch.float64, device=dev, generator=g) - 1)).contiguous()
    if by_distance:
        idx = ((xk - xi[:, None, :]) ** 2).sum(-1).argsort(dim=1)
        xk = torch.gather(xk, 1, idx[..., None].expand(-1, -1, 3)).contiguous()
    fk = (torch.sin(3 * xk[..., 0]) * torch.cos(2 * xk[..., 1]) * torch.exp(xk[..., 2])).contiguous()
    nk = torch.full((n,), K, dtype=torch.int32, device=dev); kn = torch.zeros(n, dtype=torch.int64, device=dev)
    wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
    fi = torch.zeros((n, no), dtype=torch.float64, device=dev)
    run = lambda: whip.fit_many_device(3, order, xk, fk, nk, xi, fi, kn, wm)
    run(); name = whip.last_kernel(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("3D order %d K %d n %d [%s]: %.3f ms = %.3g fits/s" % (order, K, n, name, ms, n / ms * 1e3), flush=True)
