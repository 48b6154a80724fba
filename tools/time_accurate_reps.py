import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "python-wlsqm_amd")
import numpy as np, torch, bench
import wlsqm.hip as whip
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["C2"]; n = 1_000_000
S, F, hoods = bench.build_problem(cfg, n, 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
S_d, F_d, h_d = t(S), t(F), t(hoods.astype(np.int64))
xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous(); xi = S_d.clone()
nk_d = torch.full((n,), 32, dtype=torch.int32, device=dev); wm_d = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
kn_d = torch.zeros((n,), dtype=torch.int64, device=dev)
fi = torch.zeros((n, 6), dtype=torch.float64, device=dev); fi[:, 0] = F_d
args = (2, 2, xk, fk, nk_d, xi, fi, kn_d, wm_d)
for rnd in range(3):
    for mode, ctx in (("fast", None), ("accurate", whip.accurate)):
        out = []
        for reps in (5, 20, 100):
            if ctx:
                with ctx(): ms = whip.time_fit_device(*args, reps=reps)
            else: ms = whip.time_fit_device(*args, reps=reps)
            out.append("%d: %.4f" % (reps, ms))
        # python loop with torch events
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        def run():
            whip.fit_many_device(*args, strict=("accurate" if ctx else False))
        run(); torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append("py20: %.4f" % (e0.elapsed_time(e1) / 20))
        print(rnd, mode, "  ".join(out), flush=True)
