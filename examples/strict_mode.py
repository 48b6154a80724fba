#!/usr/bin/env python3
"""The two numerics modes on one batch.

FAST (default): MI355X-first arithmetic — moment form, neighbour sums split over lanes, FMA, unpivoted LDL^T.
STRICT: every floating-point operation of the reference (make_c / weights / make_A / Ruiz scaling / pivoted LU / solve) in the
reference's own order, so that the result differs from python-wlsqm's only by LAPACK's internal summation order.
Both are as accurate against the exact derivatives; they differ from each other in the last digits, which is what this prints.
Run on a machine with one MI355X:  python examples/strict_mode.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import wlsqm
import wlsqm.hip as whip

rng = np.random.default_rng(0)
n, nk = 20000, 32
xi = rng.uniform(0.2, 0.8, (n, 2))
xk = xi[:, None, :] + 0.01 * rng.uniform(-1, 1, (n, nk, 2))                      # small neighbourhoods: second derivatives are hard
f = lambda x, y: np.sin(np.pi * x) * np.cos(np.pi * y)
fk = f(xk[..., 0], xk[..., 1])
args = dict(xk=xk, fk=fk, nk=np.full(n, nk, np.int32), xi=xi, sens=None, do_sens=0, order=np.full(n, 2, np.int32),
            knowns=np.zeros(n, np.int64), weighting_method=np.full(n, wlsqm.WEIGHT_CENTER, np.int32))
exact_xx = -np.pi ** 2 * f(xi[:, 0], xi[:, 1])

fi_fast = np.zeros((n, 6))
wlsqm.fit_2D_many_parallel(fi=fi_fast, **args)
print("fast   kernel:", whip.last_kernel())

fi_strict = np.zeros((n, 6))
with whip.strict():                                       # or WLSQM_HIP_STRICT=1 in the environment, or whip.set_strict(True)
    wlsqm.fit_2D_many_parallel(fi=fi_strict, **args)
    print("strict kernel:", whip.last_kernel())

scale = np.abs(fi_strict).max(axis=0)
print("largest difference between the modes per DOF column (relative to the column's largest value):")
print("   ", np.array2string(np.abs(fi_fast - fi_strict).max(axis=0) / scale, precision=1))
for name, fi in (("fast", fi_fast), ("strict", fi_strict)):
    print("%-6s d2f/dx2: max error against the exact derivative %.2e (truncation error of the order-2 model dominates)"
          % (name, np.abs(fi[:, wlsqm.i2_X2] - exact_xx).max()))
