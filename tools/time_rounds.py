#!/usr/bin/env python3
"""Iterative refinement of the BASELINE shapes in one launch (default) against the refinement in rounds (WLSQM_HIP_REFINE_ROUNDS=1):
ms per launch (max_iter 10), the iteration count returned, and whether the refined fi are bit-identical.
usage: python tools/time_rounds.py [ncases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
for name in ("C2", "C5"):
    cfg = bench.CONFIGS[name]
    dim, order, nk = cfg["dim"], cfg["order"], cfg["nk"]
    S, F, hoods = bench.build_problem(cfg, n, 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    S_d, F_d, h_d = t(S), t(F), t(hoods.astype(np.int64))
    xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous(); xi = S_d.clone()
    no = whip._ndofs(dim, order)
    nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev)
    wm_d = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
    kn_d = torch.full((n,), cfg["knowns"], dtype=torch.int64, device=dev)
    out = {}
    for mode in ("0", "1"):
        os.environ["WLSQM_HIP_REFINE_ROUNDS"] = mode
        fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
        def run(want=False):
            fi.zero_(); fi[:, 0] = F_d
            return whip.fit_many_device(dim, order, xk, fk, nk_d, xi, fi, kn_d, wm_d, iterative=True, max_iter=10, want_iterations=want)
        its = run(True); torch.cuda.synchronize()
        out[mode] = fi.clone()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 5
        # (the refill of fi is part of the timed region in both modes: ~0.02 ms)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        print("%s + refinement, %d cases, %s: %.3f ms per launch, iterations %d (%s)" % (name, n, "one launch" if mode == "0" else "rounds", e0.elapsed_time(e1) / reps, its, whip.last_kernel()), flush=True)
    same = torch.equal(out["0"], out["1"])
    print("   refined fi bit-identical between the two: %s" % same, flush=True)
