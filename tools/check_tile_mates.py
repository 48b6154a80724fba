#!/usr/bin/env python
"""Does a case's result depend on the other cases of its batch?  Fits mixed-knowns batches twice: as given and permuted (and through a
per-case order tensor, twice), and counts the cases whose bits differ.  python tools/check_tile_mates.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-wlsqm_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import wlsqm
import wlsqm.hip as whip
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
rng = np.random.default_rng(3)
for dim, order, Kn, masks in ((2, 4, 64, [0, 1, 1, 1, 5]), (2, 4, 40, [0, 1]), (3, 2, 40, [0, 0, 1]), (2, 2, 32, [0, 1, 2]), (2, 3, 30, [0, 1])):
    n = 4000
    no = NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim)); xk = xi[:, None, :] + 0.05 * rng.uniform(-1, 1, (n, Kn, dim))
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
    nk = np.full(n, Kn, np.int32)
    kn = rng.choice(np.array(masks, np.int64), n); wm = np.full(n, 2, np.int32)
    fi0 = rng.uniform(-1, 1, (n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
    def run(perm):
        fi = t(fi0[perm])
        whip.fit_many_device(dim, order, t(xk[perm]), t(fk[perm]), t(nk[perm]), t(xi[perm]), fi, t(kn[perm]), t(wm[perm]))
        torch.cuda.synchronize()
        out = np.empty_like(fi0); out[perm] = fi.cpu().numpy()
        return out, whip.last_kernel()
    a, k1 = run(np.arange(n)); b, _ = run(rng.permutation(n)); c, _ = run(np.arange(n))
    d1 = (a.view(np.int64) != b.view(np.int64)).any(axis=1); d2 = (a.view(np.int64) != c.view(np.int64)).any(axis=1)
    rel = np.abs(a - b).max() / np.abs(a).max()
    print("dim %d order %d K %d (%s): permuted: %d of %d cases differ (largest difference %.1e of the largest DOF), same order again: %d differ"
          % (dim, order, Kn, k1, d1.sum(), n, rel, d2.sum()), flush=True)
    # per-case order tensor twice
    orders = rng.choice(np.array([max(order - 1, 0), order], np.int32), n)
    outs = []
    for _ in range(2):
        fi = t(fi0)
        whip.fit_many_device(dim, t(orders), t(xk), t(fk), t(nk), t(xi), fi, t(kn), t(wm), max_order=order)
        torch.cuda.synchronize(); outs.append(fi.cpu().numpy())
    print("   order tensor, two runs: %d cases differ" % (outs[0].view(np.int64) != outs[1].view(np.int64)).any(axis=1).sum(), flush=True)

# the F-known shortcut of the ring kernels: a wave whose 64 cases ALL have exactly the function value known solves the reduced
# (no - 1) system directly; a mixed wave eliminates the known row of the full system.  The same case in both situations:
for dim, order, Kn in ((2, 4, 64), (2, 4, 40), (3, 2, 40), (2, 2, 32)):
    n = 4096
    no = NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim)); xk = xi[:, None, :] + 0.05 * rng.uniform(-1, 1, (n, Kn, dim))
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
    nk = np.full(n, Kn, np.int32); wm = np.full(n, 2, np.int32)
    fi0 = rng.uniform(-1, 1, (n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
    res = []
    for mixed in (False, True):
        kn = np.ones(n, np.int64)
        if mixed: kn[::7] = 0
        fi = t(fi0)
        whip.fit_many_device(dim, order, t(xk), t(fk), t(nk), t(xi), fi, t(kn), t(wm))
        torch.cuda.synchronize(); res.append(fi.cpu().numpy())
    same = np.ones(n, bool); same[::7] = False
    d = (res[0].view(np.int64) != res[1].view(np.int64)).any(axis=1) & same
    rel = np.abs(res[0][same] - res[1][same]).max(axis=0) / np.abs(res[0][same]).max(axis=0)
    print("dim %d order %d K %d (%s): F-known cases in all-F-known waves vs in mixed waves: %d of %d differ; largest relative column difference %.1e"
          % (dim, order, Kn, whip.last_kernel(), d.sum(), same.sum(), rel.max()), flush=True)
