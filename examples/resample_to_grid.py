#!/usr/bin/env python3
"""Scattered data -> local models (ExpertSolver) -> values and derivatives on a regular grid."""
import os, sys
import numpy as np
import scipy.spatial
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-wlsqm_amd"))
import wlsqm

rng = np.random.default_rng(7)
npoints, nk, order, nvis = 5000, 24, 2, 101
x = rng.uniform(0.0, 1.0, (npoints, 2))
F = np.sin(2.0 * x[:, 0]) * np.exp(-x[:, 1])

tree = scipy.spatial.cKDTree(x)
hoods = tree.query(x, 1 + nk)[1][:, 1:].astype(np.int32)     # nk nearest neighbours, the point itself dropped

solver = wlsqm.ExpertSolver(dimension=2, nk=np.full(npoints, nk, np.int32), order=np.full(npoints, order, np.int32),
                            knowns=np.full(npoints, wlsqm.b2_F, np.int64),
                            weighting_method=np.full(npoints, wlsqm.WEIGHT_UNIFORM, np.int32),
                            algorithm=wlsqm.ALGO_BASIC, do_sens=False, max_iter=10, ntasks=8, debug=False)
no = wlsqm.number_of_dofs(dimension=2, order=order)
fi = np.empty((npoints, no)); fi[:, 0] = F
solver.prepare(xi=x, xk=x[hoods])
solver.solve(fk=fi[hoods, 0], fi=fi, sens=None)

xx = np.linspace(0.05, 0.95, nvis)
X, Y = np.meshgrid(xx, xx)
grid = np.stack([X.ravel(), Y.ravel()], axis=1)
solver.prep_interpolate()                                    # index the model origins
for mode in ("nearest", "continuous"):
    kw = dict(mode=mode) if mode == "nearest" else dict(mode=mode, r=0.05)
    Z, _ = solver.interpolate(grid, **kw)
    Zx, _ = solver.interpolate(grid, diff=wlsqm.i2_X, **kw)
    print("%-10s: max |f error| = %.2e, max |df/dx error| = %.2e on a %dx%d grid"
          % (mode, np.abs(Z - np.sin(2 * grid[:, 0]) * np.exp(-grid[:, 1])).max(),
             np.abs(Zx - 2 * np.cos(2 * grid[:, 0]) * np.exp(-grid[:, 1])).max(), nvis, nvis))
