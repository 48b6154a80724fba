#!/usr/bin/env python3
"""Sensitivities / refinement of the 3D order-3/4 systems: python tools/time_extras3d.py ORDER [ncases [K]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip
order = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000; K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = dict(bench.CONFIGS["C5"], nk=K, order=order); no = bench.NDOF[3][order]
dev = torch.device("cuda", 0)
S, F, hoods = bench.build_problem(cfg, n, 0, device=dev)
S_d, F_d = torch.from_numpy(S).to(dev), torch.from_numpy(F).to(dev); h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous()
fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
nk = torch.full((n,), K, dtype=torch.int32, device=dev); kn = torch.zeros(n, dtype=torch.int64, device=dev); wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
sens = torch.zeros((n, K, no), dtype=torch.float64, device=dev)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for tag, kw in (("do_sens", dict(sens=sens)), ("iterative", dict(iterative=True, max_iter=10)), ("basic", {})):
    ms = timeit(lambda: whip.fit_many_device(3, order, xk, fk, nk, S_d, fi, kn, wm, **kw))
    print("3D order %d, K %d, %-9s: %.3f ms per %d cases -> %.3e fits/s" % (order, K, tag, ms, n, n / ms * 1e3))
