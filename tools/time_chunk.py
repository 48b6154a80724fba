#!/usr/bin/env python3
"""Chunked any-K kernel (csrc/fit_chunk.hip) against the lane-per-case kernel at large neighbour counts (200k cases).
usage: python tools/time_chunk.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import numpy as np, torch
import wlsqm.hip as whip
dev = "cuda:0"
for dim, order, K in ((2, 2, 256), (3, 2, 160), (2, 2, 130), (1, 2, 100), (2, 0, 80)):
    n = 200000
    no = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}[dim][order]
    xi = torch.rand((n, dim), dtype=torch.float64, device=dev)
    xk = (xi[:, None, :] + 0.05 * (torch.rand((n, K, dim), dtype=torch.float64, device=dev) - 0.5)).contiguous()
    fk = torch.sin(xk[..., 0]).contiguous()
    if dim == 1: xk = xk[..., 0].contiguous(); xi = xi[:, 0].contiguous()
    fi = torch.zeros((n, no), dtype=torch.float64, device=dev)
    nk = torch.full((n,), K, dtype=torch.int32, device=dev); kn = torch.zeros(n, dtype=torch.int64, device=dev); wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
    args = (dim, order, xk, fk, nk, xi, fi, kn, wm)
    out = {}
    for v in ("chunk", "lane"):
        if v == "lane": os.environ["WLSQM_HIP_DISABLE_TILE"] = "1"
        else: os.environ.pop("WLSQM_HIP_DISABLE_TILE", None)
        ms = whip.time_fit_device(*args, reps=10); out[v] = (ms, whip.last_kernel())
    os.environ.pop("WLSQM_HIP_DISABLE_TILE", None)
    B = 8 * K * (dim + 1) + 8 * dim + 8 * no + 20
    print("dim %d order %d K %d: %s %.3f ms (%.0f GB/s) | %s %.3f ms" % (dim, order, K, out["chunk"][1], out["chunk"][0], B * n / out["chunk"][0] / 1e6, out["lane"][1], out["lane"][0]))
