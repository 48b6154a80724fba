#!/usr/bin/env python3
"""Index-based path timing: python tools/tune_cloud.py CONFIG npoints [sorted|unsorted]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench, synth
import wlsqm.hip as whip
cfg = bench.CONFIGS[sys.argv[1]]; n = int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else "sorted"
dim, order, nk = cfg["dim"], int(os.environ.get("TUNE_ORDER", cfg["order"])), int(os.environ.get("TUNE_NK", cfg["nk"])); no = bench.NDOF[dim][order]
dev = torch.device("cuda", 0)
S = synth.halton(n, dim)
if mode == "sorted":
    S = np.ascontiguousarray(S[synth.morton_order(S)])
F = synth.field(S); hoods = synth.knn(S, nk)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
S_d, F_d, h_d = t(S), t(F), t(hoods)
fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev)
kn = torch.full((n,), cfg["knowns"], dtype=torch.int64, device=dev); wm = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
ms = [whip.time_fit_cloud_device(dim, order, S_d, F_d, h_d, fi, nk_d, kn, wm, reps=20) for _ in range(5)]
B_idx = 4 * nk + 8 * (dim + 1) + 8 * no + 20
xk = S_d[h_d.long()].contiguous(); fk = F_d[h_d.long()].contiguous()
fi2 = torch.zeros_like(fi); fi2[:, 0] = F_d
ms_d = [whip.time_fit_device(dim, order, xk, fk, nk_d, S_d, fi2, kn, wm, reps=20) for _ in range(5)]
print("%s %s points: index-based median %.4f ms -> %.3e fits/s (%d B/fit algorithmic -> %.0f GB/s); dense %.4f ms -> %.3e fits/s; equal=%s"
      % (sys.argv[1], mode, np.median(ms), n / np.median(ms) * 1e3, B_idx, B_idx * n / np.median(ms) / 1e6,
         np.median(ms_d), n / np.median(ms_d) * 1e3, bool(torch.equal(fi, fi2))))
