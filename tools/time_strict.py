#!/usr/bin/env python
"""Strict-mode kernel times on BASELINE shapes (ms per launch, HIP events): python tools/time_strict.py [ncases]
Prints one line per (config, extras) with the kernel family that ran; WLSQM_HIP_STRICT_NO_ROWS=1 gives the LDS kernel for A/B."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    dev = torch.device("cuda", 0)
    for name in ("C1", "C2", "C3", "C5"):
        cfg = bench.CONFIGS[name]
        dim, order, nk = cfg["dim"], cfg["order"], cfg["nk"]
        S, F, hoods = bench.build_problem(cfg, n, 0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        S_d, F_d, h_d = t(S), t(F), t(hoods.astype(np.int64))
        xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous(); xi = S_d.clone()
        from wlsqm.hip import _ndofs
        no = _ndofs(dim, order)
        nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev)
        wm_d = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
        for knowns in sorted({cfg["knowns"], 0, 1}):
            kn_d = torch.full((n,), knowns, dtype=torch.int64, device=dev)
            for mode in ("basic", "sens", "iter"):
                fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
                sens = torch.zeros((n, nk, no), dtype=torch.float64, device=dev) if mode == "sens" else None
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                def run():
                    whip.fit_many_device(dim, order, xk, fk, nk_d, xi, fi, kn_d, wm_d, sens=sens, iterative=(mode == "iter"),
                                         max_iter=10, strict=True)
                run(); torch.cuda.synchronize()
                ev0.record(); run(); run(); ev1.record(); torch.cuda.synchronize()
                print("%s knowns=%d %-5s n=%d: %8.3f ms  (%s)" % (name, knowns, mode, n, ev0.elapsed_time(ev1) / 2, whip.last_kernel()), flush=True)

if __name__ == "__main__":
    main()
