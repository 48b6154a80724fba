#!/usr/bin/env python3
"""Error of the refined fit against the extended-precision truth: refinement on the stored inverse (csrc/fit_sens.hip), the generic
kernels, the CPU oracle; and the basic fit for scale.  usage: python tools/refine_accuracy.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-wlsqm_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch, wlsqm.hip as whip
from oracle import oracle
import _parity as P
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
for dim, order, K, h in ((3, 4, 100, 0.08), (3, 4, 60, 0.3), (3, 3, 60, 0.08), (3, 3, 30, 0.01), (2, 2, 160, 0.08), (1, 2, 100, 0.08), (1, 4, 140, 0.5), (2, 0, 90, 0.1), (3, 2, 150, 0.02)):
    rng = np.random.default_rng(K); n = 300; no = NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim)); xk = xi[:, None, :] + h * rng.uniform(-1, 1, (n, K, dim))
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1]) + 0.3 * xk[..., 0]
    nk = np.full(n, K, np.int32); kn = np.zeros(n, np.int64); wm = np.full(n, 2, np.int32); orders = np.full(n, order, np.int32)
    fi0 = np.zeros((n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1]) + 0.3 * xi[:, 0]
    xk_a, xi_a = (np.ascontiguousarray(xk[..., 0]), np.ascontiguousarray(xi[:, 0])) if dim == 1 else (xk, xi)
    truth = P.truth_fit(dim, xk_a, fk, nk, xi_a, fi0, orders, kn, wm)
    res = {}
    for tag in ("new", "generic"):
        if tag == "generic": os.environ["WLSQM_HIP_DISABLE_SENS_APPLY"] = "1"
        fi = t(fi0); its = whip.fit_many_device(dim, order, t(xk_a), t(fk), t(nk), t(xi_a), fi, t(kn), t(wm), iterative=True, max_iter=10, want_iterations=True)
        res[tag] = (fi.cpu().numpy(), its, whip.last_kernel())
        os.environ.pop("WLSQM_HIP_DISABLE_SENS_APPLY", None)
    fo = fi0.copy(); ito = oracle.fit_many(dim, xk_a, fk, nk, xi_a, fo, None, 0, orders, kn, wm, iterative=True, max_iter=10)
    fb = t(fi0); whip.fit_many_device(dim, order, t(xk_a), t(fk), t(nk), t(xi_a), fb, t(kn), t(wm)); fb = fb.cpu().numpy()
    m = lambda a: float(P.column_metric(a, truth).max())
    print("dim %d order %d K %3d h %.2f: error vs truth  basic %.1e | refined: new %.1e (%s, %d it)  generic %.1e (%s, %d it)  oracle %.1e (%s it)"
          % (dim, order, K, h, m(fb), m(res["new"][0]), res["new"][2], res["new"][1], m(res["generic"][0]), res["generic"][2], res["generic"][1], m(fo), ito), flush=True)
