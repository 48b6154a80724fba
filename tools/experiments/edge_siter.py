#!/usr/bin/env python3
"""Edge shapes of the one-lane-per-case refinement kernel against the oracle: very long neighbour lists, tiny batches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import wlsqm, wlsqm.hip as whip
from oracle import oracle
import _cases as K, _parity as P
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
for dim, order, Kn, n in [(2, 4, 1000, 300), (3, 2, 500, 130), (2, 3, 2048, 70), (2, 2, 4096, 65), (2, 2, 8, 5), (3, 2, 14, 64), (2, 4, 20, 1)]:
    rng = np.random.default_rng(Kn + n)
    no = K.NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim))
    xk = xi[:, None, :] + 0.05 * rng.uniform(-1, 1, (n, Kn, dim))
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
    nk = rng.integers(min(Kn, max(no + 4, Kn // 2)), Kn + 1, n).astype(np.int32); nk[0] = Kn
    if Kn < no + 4: nk[:] = Kn
    kn = rng.choice(np.array([0, 1], np.int64), n); wm = rng.choice(np.array([1, 2], np.int32), n)
    fi0 = np.zeros((n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
    orders = np.full(n, order, np.int32)
    os.environ["WLSQM_HIP_STAGE_REFINE"] = "all"
    fi = _t(fi0)
    it = whip.fit_many_device(dim, order, _t(xk), _t(fk), _t(nk), _t(xi), fi, _t(kn), _t(wm), iterative=True, max_iter=6, want_iterations=True)
    name = whip.last_kernel(); got = fi.cpu().numpy()
    ref = fi0.copy()
    it_o = oracle.fit_many(dim, xk, fk, nk, xi, ref, None, 0, orders, kn, wm, iterative=True, max_iter=6, ntasks=8)
    E = P.column_metric(got, ref)
    truth = P.truth_fit(dim, xk, fk, nk, xi, fi0, orders, kn, wm); N = P.column_metric(ref, truth)
    ok = bool(np.all(E <= 1e-10 + 200.0 * N)) or Kn < no + 4
    print("dim %d order %d K %d n %d [%s] iters %d / %d  E_max %.2e  N_max %.2e  %s" % (dim, order, Kn, n, name, it, it_o, E.max(), N.max(), "ok" if ok else "FAIL"), flush=True)
