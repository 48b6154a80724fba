#!/usr/bin/env python3
"""Heterogeneous orders on device-resident data: two order buckets addressed through index lists (what ExpertSolver and the
host entry points do for mixed `order` arrays).  python tools/time_buckets.py [ncases]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg = bench.CONFIGS["C2"]; dim, nk = 2, 32
dev = torch.device("cuda", 0)
S, F, hoods = bench.build_problem(cfg, n, 0, device=dev)
S_d, F_d = torch.from_numpy(S).to(dev), torch.from_numpy(F).to(dev); h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous()
fi = torch.zeros((n, 6), dtype=torch.float64, device=dev); fi[:, 0] = F_d
nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev); kn = torch.zeros(n, dtype=torch.int64, device=dev)
wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
idx = torch.arange(n, device=dev)
b1, b2 = idx[idx % 3 == 0].contiguous(), idx[idx % 3 != 0].contiguous()        # order 1 for a third of the cases, order 2 for the rest
def run():
    whip.fit_many_device(dim, 1, xk, fk, nk_d, S_d, fi, kn, wm, case_index=b1)
    whip.fit_many_device(dim, 2, xk, fk, nk_d, S_d, fi, kn, wm, case_index=b2)
for tag in ("tile", "lane"):
    if tag == "lane": os.environ["WLSQM_HIP_DISABLE_TILE"] = "1"
    run(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print("%s kernels: %.3f ms per %d mixed-order cases -> %.3e fits/s" % (tag, ms, n, n / ms * 1e3))
