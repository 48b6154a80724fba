#!/usr/bin/env python3
"""Time the iterative refinement per (dimension, order, K) shape and max_iter: the one-lane-per-case kernel (round 4) against the kernels
the shape took before and against the lane kernel (TIME_REFINE_MODES=stage,before,lane).
usage: python tools/time_refine.py [ncases] [dim,order,K ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import wlsqm.hip as whip
import synth
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
dev = torch.device("cuda", 0)
args = sys.argv[1:]
n = int(args[0]) if args and "," not in args[0] else 400000
shapes = [tuple(int(v) for v in a.split(",")) for a in args if "," in a] or [(2, 4, 64), (2, 3, 80), (2, 4, 30)]
for dim, order, K in shapes:
    no = NDOF[dim][order]
    S = torch.from_numpy(synth.halton(n, dim)).to(dev)
    F = torch.from_numpy(synth.field(S.cpu().numpy())).to(dev)
    h = whip.knn(S, K).long()
    xk = S[h].contiguous(); fk = F[h].contiguous()
    nk = torch.full((n,), K, dtype=torch.int32, device=dev)
    kn = torch.full((n,), 1, dtype=torch.int64, device=dev)
    wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
    fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F
    for mode in os.environ.get("TIME_REFINE_MODES", "stage,before,lane").split(","):      # stage: csrc/fit_stage_iter.hip (round 4); before: tile1-extras / chunk-refine
        os.environ.pop("WLSQM_HIP_DISABLE_CHUNK_REFINE", None); os.environ.pop("WLSQM_HIP_STAGE_REFINE", None)
        if mode != "stage":
            os.environ["WLSQM_HIP_STAGE_REFINE"] = "0"
        elif os.environ.get("TIME_REFINE_FORCE") == "1":          # every covered shape on the kernel, whatever the dispatch rule says
            os.environ["WLSQM_HIP_STAGE_REFINE"] = "all"
        if mode == "lane":
            os.environ["WLSQM_HIP_DISABLE_CHUNK_REFINE"] = "1"
        row = []
        for mi in (0, 1, 2, 4, 10):
            run = lambda: whip.fit_many_device(dim, order, xk, fk, nk, S, fi, kn, wm, iterative=True, max_iter=mi)
            run(); name = whip.last_kernel(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            row.append("max_iter %d: %.3f ms" % (mi, e0.elapsed_time(e1) / 5))
        print("dim %d order %d K %d n %d [%s -> %s]  %s" % (dim, order, K, n, mode, name, "  ".join(row)), flush=True)
