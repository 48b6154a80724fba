#!/usr/bin/env python3
"""BASELINE config 4 (ExpertSolver: prepare once + many right-hand sides) on one GPU, device-resident.
usage: python tools/time_solve_many.py [ncases [nrhs_per_call [calls]]]
Prints the time-stepping rate (one solve_device launch per field) and the stacked rate (solve_many_device)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = bench.CONFIGS["C2"]; dim, order, nk, no = 2, 2, 32, 6
dev = torch.device("cuda", 0)
S, F, hoods = bench.build_problem(cfg, n, 0)
s = wlsqm.ExpertSolver(dimension=dim, nk=np.full(n, nk, np.int32), order=np.full(n, order, np.int32),
                       knowns=np.zeros(n, np.int64), weighting_method=np.full(n, cfg["wm"], np.int32))
t0 = time.perf_counter(); s.prepare(xi=S, xk=S[hoods]); t_prep = time.perf_counter() - t0
S_d = torch.from_numpy(S).to(dev); h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
fk = torch.empty((R, n, nk), dtype=torch.float64, device=dev)
for r in range(R):
    Ft = torch.sin(np.pi * S_d[:, 0] + 0.01 * r) * torch.cos(np.pi * S_d[:, 1])
    fk[r] = Ft[h_d]
fi = torch.zeros((R, n, no), dtype=torch.float64, device=dev)
fi_seq = torch.zeros_like(fi)
def timed(f, reps):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
def seq():
    for r in range(R): s.solve_device(fk[r], fi_seq[r])
t_seq = timed(seq, calls)
t_many = timed(lambda: s.solve_many_device(fk, fi), calls)
err = float(((fi - fi_seq).abs().amax(dim=(0, 1)) / fi_seq.abs().amax(dim=(0, 1))).max())
B_seq = bench.bytes_per_fit(dim, order, nk, 0)
B_many = 8 * nk + 8 * no + (8 * nk * dim + 8 * dim + 20) / R
print("prepare (host arrays -> HBM): %.1f ms" % (t_prep * 1e3))
print("time stepping, %d x solve_device : %.3f ms per field -> %.3e fits/s (%d B/fit -> %.0f GB/s)"
      % (R, t_seq / R * 1e3, n * R / t_seq, B_seq, B_seq * n * R / t_seq / 1e9))
print("stacked, solve_many_device(%d)   : %.3f ms per field -> %.3e fits/s (%.0f B/fit -> %.0f GB/s, %.1f%% of 8 TB/s); max col rel diff vs sequential %.1e"
      % (R, t_many / R * 1e3, n * R / t_many, B_many, B_many * n * R / t_many / 1e9, B_many * n * R / t_many / 1e9 / 80, err))
