#!/usr/bin/env python3
"""The one-lane-per-case staged kernel (csrc/fit_stage.hip, WLSQM_HIP_STAGE=all) against the kernels the dispatcher picks without
it (WLSQM_HIP_STAGE=0), per shape and neighbour count (multiples of 8), same box, interleaved: ms per 400k-case launch.
usage: python tools/sweep_stage.py [ncases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import wlsqm.hip as whip
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu"); g.manual_seed(0)
for dim, order, kn, Ks in ((2, 4, 1, range(24, 105, 8)), (3, 2, 0, range(16, 129, 8)), (2, 3, 1, range(16, 129, 16)), (2, 2, 0, range(8, 129, 8))):
    no = NDOF[dim][order]
    for K in Ks:
        xi = torch.rand((n, dim), dtype=torch.float64, generator=g).to(dev)
        off = (torch.rand((n, K, dim), dtype=torch.float64, generator=g).to(dev) - 0.5) * 0.1
        # sorted by distance (what a k-nearest-neighbour search returns): the staged kernel's speculation holds
        d2 = (off * off).sum(dim=2); idx = d2.argsort(dim=1)
        off = torch.gather(off, 1, idx[:, :, None].expand(-1, -1, dim))
        xk = (xi[:, None, :] + off).contiguous()
        fk = (torch.sin(3 * xk[..., 0]) * torch.cos(2 * xk[..., -1])).contiguous()
        nk = torch.full((n,), K, dtype=torch.int32, device=dev); wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
        knd = torch.full((n,), kn, dtype=torch.int64, device=dev)
        fi = torch.zeros((n, no), dtype=torch.float64, device=dev)
        res = {}
        for rep in range(2):
            for mode in ("0", "all"):
                os.environ["WLSQM_HIP_STAGE"] = mode
                run = lambda: whip.fit_many_device(dim, order, xk, fk, nk, xi, fi, knd, wm)
                run(); torch.cuda.synchronize(); name = whip.last_kernel()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): run()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(mode, []).append((e0.elapsed_time(e1) / 5, name))
        a = min(v[0] for v in res["0"]); b = min(v[0] for v in res["all"])
        print("dim %d order %d K %3d: %-12s %.3f ms   stage %.3f ms   ratio %.2f" % (dim, order, K, res["0"][0][1], a, b, a / b), flush=True)
