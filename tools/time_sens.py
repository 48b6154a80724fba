#!/usr/bin/env python3
"""Device-resident timing of the sensitivities / iterative paths: python tools/time_sens.py [ncases [lane]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench, synth
import wlsqm.hip as whip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg = dict(bench.CONFIGS[os.environ.get("TUNE_CONFIG", "C2")]); cfg["nk"] = int(os.environ.get("TUNE_NK", cfg["nk"]))
cfg["order"] = int(os.environ.get("TUNE_ORDER", cfg["order"]))
dim, order, nk = cfg["dim"], cfg["order"], cfg["nk"]; no = bench.NDOF[dim][order]
dev = torch.device("cuda", 0)
S, F, hoods = bench.build_problem(cfg, n, 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
S_d, F_d, h_d = t(S), t(F), t(hoods.astype(np.int64))
xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous()
fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev); kn = torch.zeros(n, dtype=torch.int64, device=dev)
wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
sens = torch.zeros((n, nk, no), dtype=torch.float64, device=dev)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
if len(sys.argv) > 2 and sys.argv[2] == "lane":
    os.environ["WLSQM_HIP_DISABLE_TILE_EXTRAS"] = "1"
ms = timeit(lambda: whip.fit_many_device(dim, order, xk, fk, nk_d, S_d, fi, kn, wm, sens=sens))
print("do_sens  : %.3f ms -> %.3e fits/s (%s, K = %d, kernel %s)" % (ms, n / ms * 1e3, os.environ.get("TUNE_CONFIG", "C2"), nk, whip.last_kernel()))
ms = timeit(lambda: whip.fit_many_device(dim, order, xk, fk, nk_d, S_d, fi, kn, wm, iterative=True, max_iter=10))
print("iterative: %.3f ms -> %.3e fits/s" % (ms, n / ms * 1e3))
ms = timeit(lambda: whip.fit_many_device(dim, order, xk, fk, nk_d, S_d, fi, kn, wm))
print("basic    : %.3f ms -> %.3e fits/s" % (ms, n / ms * 1e3))
