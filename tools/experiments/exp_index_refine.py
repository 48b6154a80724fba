import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip
os.environ["WLSQM_HIP_REFINE_ROUNDS"] = "0"
n = 1_000_000
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
    fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
    rng = np.random.default_rng(0)
    for label, idx in (("none", None), ("identity", np.arange(n)), ("47% sorted", np.sort(rng.choice(n, int(0.47*n), replace=False))),
                       ("47% shuffled", rng.choice(n, int(0.47*n), replace=False)), ("9% sorted", np.sort(rng.choice(n, int(0.09*n), replace=False))),
                       ("9% shuffled", rng.choice(n, int(0.09*n), replace=False)), ("9% chunks of 4 shuffled", None)):
        if label.startswith("9% chunks"):
            base = rng.choice(n // 4, int(0.09*n) // 4, replace=False) * 4
            idx = (base[:, None] + np.arange(4)[None, :]).ravel()
        ci = None if idx is None else t(idx.astype(np.int64))
        for mi in (0, 3):
            run = lambda: whip.fit_many_device(dim, order, xk, fk, nk_d, xi, fi, kn_d, wm_d, iterative=True, max_iter=mi, case_index=ci)
            run(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            print("%s %-26s max_iter %d: %.3f ms (%s)" % (name, label, mi, e0.elapsed_time(e1) / 5, whip.last_kernel()), flush=True)
