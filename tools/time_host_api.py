#!/usr/bin/env python3
"""Throughput of the reference-signature (host array) API: python tools/time_host_api.py [ncases]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import synth, wlsqm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = synth.cloud_problem(2, n, 32)
nk = np.full(n, 32, np.int32); o = np.full(n, 2, np.int32); kn = np.zeros(n, np.int64); w = np.full(n, 2, np.int32)
fi = np.zeros((n, 6)); fi[:, 0] = p["F"]
for rep in range(4):
    t0 = time.perf_counter()
    wlsqm.fit_2D_many_parallel(p["xk"], p["fk"], nk, p["xi"], fi, None, 0, o, kn, w)
    dt = time.perf_counter() - t0
    print("fit_2D_many_parallel(host arrays, %d cases): %.3f s -> %.3e fits/s" % (n, dt, n / dt))
s = wlsqm.ExpertSolver(2, nk, o, kn, w)
t0 = time.perf_counter(); s.prepare(p["xi"], p["xk"]); print("ExpertSolver.prepare: %.3f s" % (time.perf_counter() - t0))
for rep in range(3):
    t0 = time.perf_counter(); s.solve(p["fk"], fi); dt = time.perf_counter() - t0
    print("ExpertSolver.solve(host fk/fi): %.3f s -> %.3e fits/s" % (dt, n / dt))
