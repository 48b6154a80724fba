import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C3", "C5"]
dev = torch.device("cuda", 0)
for name in names:
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
    run = lambda: whip.fit_many_device(dim, order, xk, fk, nk_d, xi, fi, kn_d, wm_d)
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    B = {"C2": 852, "C3": 1700, "C5": 1404}[name]
    print("%s n=%d: %.4f ms  frac %.3f  (%s)" % (name, n, ms, B * n / (ms * 1e-3) / 8e12, whip.last_kernel()), flush=True)
