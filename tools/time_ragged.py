#!/usr/bin/env python3
"""Ragged neighbourhoods (nk[j] uniform in [K/2, K]) against full ones: python tools/time_ragged.py CONFIG [ncases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip
cfg = bench.CONFIGS[sys.argv[1]]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dim, order, nk = cfg["dim"], cfg["order"], int(os.environ.get("TUNE_NK", cfg["nk"]))
cfg = dict(cfg, nk=nk); no = bench.NDOF[dim][order]
dev = torch.device("cuda", 0)
S, F, hoods = bench.build_problem(cfg, n, 0, device=dev)
S_d, F_d = torch.from_numpy(S).to(dev), torch.from_numpy(F).to(dev); h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous()
fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
kn = torch.full((n,), cfg["knowns"], dtype=torch.int64, device=dev); wm = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
full = torch.full((n,), nk, dtype=torch.int32, device=dev)
rag = torch.from_numpy(np.random.default_rng(0).integers(max(no + 1, nk // 2), nk + 1, n).astype(np.int32)).to(dev)
for name, nkd in (("full", full), ("ragged", rag)):
    ms = np.median([whip.time_fit_device(dim, order, xk, fk, nkd, S_d, fi, kn, wm, reps=20) for _ in range(5)])
    print("%s %s nk: %.4f ms per %d cases -> %.3e fits/s" % (sys.argv[1], name, ms, n, n / ms * 1e3))
