#!/usr/bin/env python3
"""Reproduce tests/test_gpu_round4.py::test_staged_refinement_kernel[sorted-1000-3-2-124] and print where the NaN patterns differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import wlsqm, wlsqm.hip as whip
from oracle import oracle
import _cases as K
dim, order, Kn, n = 3, 2, 124, 1000
rng = np.random.default_rng(17 * Kn + n + dim)
no = K.NDOF[dim][order]
xi = rng.uniform(0, 1, (n, dim))
off = 0.05 * rng.uniform(-1, 1, (n, Kn, dim))
nk = rng.integers(min(Kn, max(no + 10, Kn // 3)), Kn + 1, n).astype(np.int32); nk[::3] = Kn
for j in range(n):
    idx = np.argsort((off[j, :nk[j]] ** 2).sum(axis=1), kind="stable")
    off[j, :nk[j]] = off[j, :nk[j]][idx]
xk = xi[:, None, :] + off
fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
kn = rng.choice(np.array([0, 0, 1, 1 | (1 << (no - 1)), (1 << no) - 1, 1 << (no + 2)], np.int64), n)
wm = rng.choice(np.array([wlsqm.WEIGHT_UNIFORM, wlsqm.WEIGHT_CENTER], np.int32), n)
fi0 = rng.uniform(-1, 1, (n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
orders = np.full(n, order, np.int32)
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
def run(**env):
    os.environ.update(env)
    fi = _t(fi0)
    it = whip.fit_many_device(dim, order, _t(xk), _t(fk), _t(nk), _t(xi), fi, _t(kn), _t(wm), iterative=True, max_iter=10, want_iterations=True)
    for k in env: os.environ.pop(k)
    return fi.cpu().numpy(), it, whip.last_kernel()
got, it, name = run()
old, it_old, name_old = run(WLSQM_HIP_STAGE_REFINE="0")
ref = fi0.copy()
it_o = oracle.fit_many(dim, xk, fk, nk, xi, ref, None, 0, orders, kn, wm, iterative=True, max_iter=10, ntasks=8)
print(name, it, name_old, it_old, "oracle", it_o)
for label, a in (("new", got), ("old", old), ("oracle", ref)):
    rows = np.flatnonzero(np.isnan(a).any(axis=1))
    print(label, "NaN rows", rows[:20], "kn", kn[rows][:20], "nk", nk[rows][:20], "wm", wm[rows][:20])
bad = np.flatnonzero(np.isnan(got).any(axis=1) != np.isnan(ref).any(axis=1))
for j in bad[:5]:
    print("case", j, "lane", j % 64, "kn", kn[j], "nk", nk[j], "wm", wm[j]); print(" new", got[j]); print(" old", old[j]); print(" ora", ref[j])
