#!/usr/bin/env python3
"""Sensitivities / refinement through the inverse (csrc/fit_sens.hip) against the generic kernels at batch sizes around the 64-case groups
and slices of that path (1, 2, 15, 63, 64, 65, 129 cases).  usage (GPU box): python tools/check_sens_edges.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-wlsqm_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch, wlsqm.hip as whip
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
bad = 0
for dim, order, K in ((2, 4, 50), (2, 4, 130), (3, 3, 40), (3, 4, 70), (2, 2, 140), (1, 2, 100)):
    for n in (1, 2, 15, 63, 64, 65, 129):
        rng = np.random.default_rng(n + K); no = NDOF[dim][order]
        xi = rng.uniform(0, 1, (n, dim)); xk = xi[:, None, :] + 0.1 * rng.uniform(-1, 1, (n, K, dim))
        fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
        nk = rng.integers(K - 5, K + 1, n).astype(np.int32); kn = rng.choice(np.array([0, 1], np.int64), n); wm = np.full(n, 2, np.int32)
        fi0 = np.zeros((n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
        xk_a, xi_a = (xk[..., 0], xi[:, 0]) if dim == 1 else (xk, xi)
        res = {}
        for tag in ("new", "generic"):
            if tag == "generic": os.environ["WLSQM_HIP_DISABLE_SENS_APPLY"] = "1"
            fi = t(fi0); sens = torch.full((n, K, no), 777.0, dtype=torch.float64, device="cuda:0")
            whip.fit_many_device(dim, order, t(xk_a), t(fk), t(nk), t(xi_a), fi, t(kn), t(wm), sens=sens); ks = whip.last_kernel()
            fr = t(fi0); whip.fit_many_device(dim, order, t(xk_a), t(fk), t(nk), t(xi_a), fr, t(kn), t(wm), iterative=True, max_iter=5); kr = whip.last_kernel()
            torch.cuda.synchronize()
            res[tag] = (fi.cpu().numpy(), sens.cpu().numpy(), fr.cpu().numpy(), ks, kr)
            os.environ.pop("WLSQM_HIP_DISABLE_SENS_APPLY", None)
        a, b = res["new"], res["generic"]
        ok = np.array_equal(np.isnan(a[1]), np.isnan(b[1])) and np.array_equal(a[1] == 777.0, b[1] == 777.0)
        x, y = np.nan_to_num(a[1]), np.nan_to_num(b[1]); live = y != 777.0
        sc = np.abs(np.where(live, y, 0)).max(axis=(1, 2), keepdims=True) + 1e-300
        ds = float((np.abs(np.where(live, x - y, 0)) / sc).max())
        df = float(np.abs(a[0] - b[0]).max() / (np.abs(b[0]).max() + 1e-300)); dr = float(np.abs(a[2] - b[2]).max() / (np.abs(b[2]).max() + 1e-300))
        ok = ok and ds < 1e-5 and df < 1e-5 and dr < 1e-5 and np.isfinite(a[0]).all() and np.isfinite(a[2]).all()
        bad += not ok
        print("dim %d order %d K %3d n %3d: %s / %s vs %s / %s  sens %.1e fi %.1e refined %.1e  %s" % (dim, order, K, n, a[3], a[4], b[3], b[4], ds, df, dr, "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
