#!/usr/bin/env python
"""Fast / accurate / strict kernel times on BASELINE shapes (ms per launch, HIP events): python tools/time_accurate.py [ncases] [configs] [shuffle]
("shuffle": the neighbours of every case in random order, as a ball query delivers them.)  One line per (config, mode): ms, fraction of the 8 TB/s HBM peak from the algorithmic bytes, kernel family."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-wlsqm_amd"))
import torch
import bench
import wlsqm.hip as whip

BYTES = {"C2": 852, "C5": 1404, "C3": 1700, "C1": 180}

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C2", "C5"]
    shuffle = len(sys.argv) > 3 and sys.argv[3] == "shuffle"
    dev = torch.device("cuda", 0)
    for name in names:
        cfg = bench.CONFIGS[name]
        dim, order, nk = cfg["dim"], cfg["order"], cfg["nk"]
        S, F, hoods = bench.build_problem(cfg, n, 0)
        if shuffle:
            rng = np.random.default_rng(5)
            hoods = rng.permuted(hoods, axis=1)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        S_d, F_d, h_d = t(S), t(F), t(hoods.astype(np.int64))
        xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous(); xi = S_d.clone()
        from wlsqm.hip import _ndofs
        no = _ndofs(dim, order)
        nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev)
        wm_d = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
        kn_d = torch.full((n,), int(os.environ.get("WLSQM_TIME_KNOWNS", cfg["knowns"])), dtype=torch.int64, device=dev)
        modes = [{"fast": False, "accurate": 2, "strict": True}[m] for m in os.environ.get("WLSQM_TIME_MODES", "fast,accurate,strict").split(",")]
        for mode in modes:
            fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            def run():
                whip.fit_many_device(dim, order, xk, fk, nk_d, xi, fi, kn_d, wm_d, strict=mode)
            run(); torch.cuda.synchronize()
            reps = 5
            ev0.record()
            for _ in range(reps): run()
            ev1.record(); torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / reps
            print("%s %-8s n=%d: %8.4f ms  frac %.3f  (%s)" % (name, {False: "fast", 2: "accurate", True: "strict"}[mode], n, ms,
                                                                BYTES[name] * n / (ms * 1e-3) / 8e12, whip.last_kernel()), flush=True)

if __name__ == "__main__":
    main()
