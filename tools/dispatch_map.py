#!/usr/bin/env python3
"""Which kernel family runs for (dimension, order, K, input form, extras)?  Tiny device-resident batches, wlsqm_hip_last_kernel.
usage: python tools/dispatch_map.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import wlsqm.hip as whip
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
dev = torch.device("cuda", 0)
n = 512          # >= 256: batches this size are repacked on the device when their layout needs it
Ks = [4, 7, 8, 12, 16, 21, 30, 32, 40, 50, 64, 66, 80, 100, 124, 128, 130, 160, 256]
rng = np.random.default_rng(0)
for dim in (1, 2, 3):
    for order in range(5):
        no = NDOF[dim][order]
        row = []
        for K in Ks:
            if K < no + 1:
                row.append("-"); continue
            xi = torch.from_numpy(rng.uniform(0, 1, (n, dim))).to(dev)
            xk = (xi[:, None, :] + 0.05 * torch.from_numpy(rng.uniform(-1, 1, (n, K, dim))).to(dev)).contiguous()
            fk = torch.sin(xk[..., 0]).contiguous()
            xi_a, xk_a = (xi[:, 0].contiguous(), xk[..., 0].contiguous()) if dim == 1 else (xi, xk)
            fi = torch.zeros((n, no), dtype=torch.float64, device=dev)
            nk = torch.full((n,), K, dtype=torch.int32, device=dev); kn = torch.zeros(n, dtype=torch.int64, device=dev)
            wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
            names = []
            whip.fit_many_device(dim, order, xk_a, fk, nk, xi_a, fi, kn, wm); names.append(whip.last_kernel())
            sens = torch.zeros((n, K, no), dtype=torch.float64, device=dev)
            whip.fit_many_device(dim, order, xk_a, fk, nk, xi_a, fi, kn, wm, sens=sens); names.append(whip.last_kernel())
            whip.fit_many_device(dim, order, xk_a, fk, nk, xi_a, fi, kn, wm, iterative=True, max_iter=3); names.append(whip.last_kernel())
            # simply-strided dense input: the neighbour axis of xk and fk has a stride (views into wider arrays)
            if dim == 1:
                xk_s = torch.zeros((n, K, 2), dtype=torch.float64, device=dev)[:, :, 0]; xk_s.copy_(xk_a)
            else:
                xk_s = torch.zeros((n, K + 3, dim), dtype=torch.float64, device=dev)[:, :K]; xk_s.copy_(xk_a)
            fk_s = torch.zeros((n, K, 2), dtype=torch.float64, device=dev)[:, :, 0]; fk_s.copy_(fk)
            whip.fit_many_device(dim, order, xk_s, fk_s, nk, xi_a, fi, kn, wm); names.append(whip.last_kernel())
            S = torch.cat([xk.reshape(n * K, dim), xi]).contiguous(); F = torch.cat([fk.reshape(n * K), torch.sin(xi[:, 0])]).contiguous()
            S_a = S[:, 0].contiguous() if dim == 1 else S
            hoods = torch.arange(n * K, dtype=torch.int32, device=dev).reshape(n, K).contiguous()
            pidx = (n * K + torch.arange(n, device=dev)).to(torch.int32)
            whip.fit_cloud_device(dim, order, S_a, F, hoods, fi, nk, kn, wm, point_index=pidx); names.append(whip.last_kernel())
            row.append("/".join(names))
        print("dim %d order %d:" % (dim, order))
        for k, r in zip(Ks, row):
            print("    K=%d %s" % (k, r))
torch.cuda.synchronize()
print("(dense basic / dense with sensitivities / dense iterative / strided dense basic / index-based basic)")
