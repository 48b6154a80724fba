#!/usr/bin/env python3
"""GPU neighbour search vs scipy cKDTree: python tools/time_knn.py [npoints [dim [k]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import synth
import wlsqm.hip as whip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 2
k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
S = synth.halton(n, dim)
S_d = torch.from_numpy(S).cuda()
whip.knn(S_d[:10000].contiguous(), k)
torch.cuda.synchronize(); t0 = time.perf_counter()
h = whip.knn(S_d, k)
torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); ref = synth.knn(S, k, workers=-1); t_cpu = time.perf_counter() - t0
same = float((h.cpu().numpy() == ref).all(axis=1).mean())
print("%d points, dim %d, k %d: GPU grid search %.1f ms (%.2e queries/s); cKDTree with %d threads %.1f ms; identical rows %.4f"
      % (n, dim, k, t_gpu * 1e3, n / t_gpu, len(os.sched_getaffinity(0)), t_cpu * 1e3, same))
