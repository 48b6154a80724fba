#!/usr/bin/env python
"""Index-based (S, F, hoods) against dense (xk, fk) input of the same fits, ms per launch (HIP events), points in Morton order
(TIME_CLOUD_MORTON=0: Halton order, every gather a cache miss):
python tools/time_cloud.py [ncases] [dim.order.K,...]   — default: 2D order 4 at K = 32 / 48 / 64 (F known, C3's mask) and 3D order 2 at K = 40."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip
from wlsqm.hip import _ndofs

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device("cuda", 0)
    shapes = [("C3", 32, None), ("C3", 48, None), ("C3", 64, None), ("C5", 40, None)]
    if len(sys.argv) > 2:                                     # "dim.order.K,..." e.g. 3.3.40,3.4.64,2.2.24
        shapes = []
        for tok in sys.argv[2].split(","):
            d, o, K = (int(x) for x in tok.split("."))
            shapes.append(("C5" if d == 3 else "C3" if o == 4 else "C2", K, o))
    for name, K, order_override in shapes:
        cfg = dict(bench.CONFIGS[name]); cfg["nk"] = K
        if order_override is not None: cfg["order"] = order_override
        dim, order = cfg["dim"], cfg["order"]
        S, F, hoods = bench.build_problem(cfg, n, 0)
        if os.environ.get("TIME_CLOUD_MORTON", "1") == "1":      # points along a space-filling curve: the gathers of a tile hit L2
            import synth
            perm = synth.morton_order(S); inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
            S, F, hoods = S[perm], F[perm], inv[hoods[perm]]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        S_d, F_d = t(S), t(F)
        h32 = t(hoods.astype(np.int32)); h64 = h32.long()
        xk = S_d[h64].contiguous(); fk = F_d[h64].contiguous(); xi = S_d.clone()
        no = _ndofs(dim, order)
        nk_d = torch.full((n,), K, dtype=torch.int32, device=dev)
        wm_d = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
        kn_d = torch.full((n,), cfg["knowns"], dtype=torch.int64, device=dev)
        fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
        fi2 = fi.clone()
        whip.time_fit_device(dim, order, xk, fk, nk_d, xi, fi, kn_d, wm_d, reps=5)                  # warm-up
        ms_d = whip.time_fit_device(dim, order, xk, fk, nk_d, xi, fi, kn_d, wm_d, reps=40); kd = whip.last_kernel()
        whip.time_fit_cloud_device(dim, order, S_d, F_d, h32, fi2, nk_d, kn_d, wm_d, reps=5)
        ms_c = whip.time_fit_cloud_device(dim, order, S_d, F_d, h32, fi2, nk_d, kn_d, wm_d, reps=40); kc = whip.last_kernel()
        same = float((fi - fi2).abs().max())
        print("%dD order %d K=%d n=%d: dense %.4f ms (%s), index-based %.4f ms (%s), max |dense - indexed| %.2e" %
              (dim, order, K, n, ms_d, kd, ms_c, kc, same), flush=True)

if __name__ == "__main__":
    main()
