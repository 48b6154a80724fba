#!/usr/bin/env python3
"""Bitwise repeatability of the sensitivities / refinement paths: 60 repetitions per shape with allocator churn in between, every output compared
with the first repetition (WLSQM_HIP_DISABLE_SENS_APPLY=1: the generic kernels).  usage (GPU box): python tools/check_determinism.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-wlsqm_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch, wlsqm.hip as whip
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
for dim, order, Kn, n in ((3, 4, 130, 40), (3, 3, 60, 120), (2, 4, 50, 300), (2, 2, 160, 260)):
    rng = np.random.default_rng(11 * Kn + order); no = NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim)); xk = xi[:, None, :] + 0.08 * rng.uniform(-1, 1, (n, Kn, dim))
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
    nk = rng.integers(min(Kn, max(no + 1, Kn - 30)), Kn + 1, n).astype(np.int32); nk[0] = Kn
    masks = [0, 0, 1, 1 << (no - 1), 1 | (1 << (no // 2)), 1 << no, 1 | (1 << (no + 2))]
    kn = rng.choice(np.array(masks, np.int64), n); wm = rng.choice(np.array([1, 2], np.int32), n)
    fi0 = rng.uniform(-1, 1, (n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
    args = (t(xk), t(fk), t(nk), t(xi))
    ref = None; diffs = 0
    for rep in range(60):
        # churn the allocator / caches a little between repetitions
        junk = torch.rand((int(rng.integers(1, 40)) * 100000,), device="cuda:0"); del junk
        fi_s = t(fi0); sens = torch.full((n, Kn, no), 777.0, dtype=torch.float64, device="cuda:0")
        whip.fit_many_device(dim, order, *args, fi_s, t(kn), t(wm), sens=sens)
        fi_r = t(fi0); its = whip.fit_many_device(dim, order, *args, fi_r, t(kn), t(wm), iterative=True, max_iter=8, want_iterations=True)
        fi_b = t(fi0); sens_b = torch.full((n, Kn, no), 777.0, dtype=torch.float64, device="cuda:0")
        whip.fit_many_device(dim, order, *args, fi_b, t(kn), t(wm), iterative=True, max_iter=8, sens=sens_b)
        torch.cuda.synchronize()
        cur = (fi_s.cpu().numpy(), sens.cpu().numpy(), fi_r.cpu().numpy(), its, fi_b.cpu().numpy(), sens_b.cpu().numpy())
        if ref is None: ref = cur
        else:
            same = all(np.array_equal(a, b, equal_nan=True) if isinstance(a, np.ndarray) else a == b for a, b in zip(ref, cur))
            same = same and np.array_equal(cur[2], cur[4]) and np.array_equal(cur[1], cur[5], equal_nan=True)
            if not same:
                diffs += 1
                which = [i for i, (a, b) in enumerate(zip(ref, cur)) if not (np.array_equal(a, b, equal_nan=True) if isinstance(a, np.ndarray) else a == b)]
                print("   rep %d differs in outputs %s; both-vs-new fi %s sens %s" % (rep, which, np.array_equal(cur[2], cur[4]), np.array_equal(cur[1], cur[5], equal_nan=True)), flush=True)
    print("dim %d order %d K %d n %d: %d of 59 repetitions differ from the first" % (dim, order, Kn, n, diffs), flush=True)
