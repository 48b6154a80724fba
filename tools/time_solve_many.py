#!/usr/bin/env python3
"""Stacked right-hand sides on one prepared geometry (BASELINE configs[3]) on one GPU, device-resident: A/B of the kernels.
usage: python tools/time_solve_many.py [GEOMETRY [ncases [nrhs [calls]]]]      GEOMETRY = C2 | C3 | C5 (BASELINE shapes)
Arms: "seq" = one fused solve_device launch per field (time stepping), "fma" = solve_many.hip (geometry shared inside the launch,
FMA loop; no <= 6, K <= 32 only), "op" = solve_op.hip (stored solution operator, v_mfma_f64_16x16x4_f64).  Every arm is compared
with the sequential result (max relative difference per DOF column) and, on the first 512 cases, with the CPU oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import wlsqm
import wlsqm.hip as whip

geo = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cfg = bench.CONFIGS[geo]; dim, order, nk = cfg["dim"], cfg["order"], cfg["nk"]; no = bench.NDOF[dim][order]
dev = torch.device("cuda", 0)
S, F, hoods = bench.build_problem(cfg, n, 0, device=dev)
s = wlsqm.ExpertSolver(dimension=dim, nk=np.full(n, nk, np.int32), order=np.full(n, order, np.int32),
                       knowns=np.full(n, cfg["knowns"], np.int64), weighting_method=np.full(n, cfg["wm"], np.int32))
S_d = torch.from_numpy(S).to(dev); h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
s.prepare_device(S_d, S_d[h_d].contiguous())
fk = torch.empty((R, n, nk), dtype=torch.float64, device=dev)
fi0 = torch.zeros((R, n, no), dtype=torch.float64, device=dev)
for r in range(R):
    Ft = torch.sin(np.pi * S_d[:, 0] + 0.01 * r) * torch.cos(np.pi * S_d[:, 1])
    if dim == 3:
        Ft = Ft * torch.exp(S_d[:, 2])
    fk[r] = Ft[h_d]; fi0[r, :, 0] = Ft
fi_seq = fi0.clone()


def timed(f, reps):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


def seq():
    for r in range(R):
        s.solve_device(fk[r], fi_seq[r])
t_seq = timed(seq, 1)
B_seq = bench.bytes_per_fit(dim, order, nk, cfg["knowns"])
B_many = 8 * nk + 8 * no + 8 * bin(cfg["knowns"]).count("1")
print("%s geometry (dim %d order %d, %d neighbours, no %d), %d cases x %d fields" % (geo, dim, order, nk, no, n, R))
print("  seq : %.3f ms per field -> %.3e fits/s (%d B/fit -> %.0f GB/s)   [%s]" % (t_seq / R * 1e3, n * R / t_seq, B_seq, B_seq * n * R / t_seq / 1e9, whip.last_kernel()))
for arm in ("fma", "op"):
    os.environ["WLSQM_HIP_SOLVE_MANY"] = arm
    fi = fi0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.solve_many_device(fk, fi); torch.cuda.synchronize()
    t_first = time.perf_counter() - t0                                 # includes the operator build of the "op" arm
    kern = whip.last_kernel()
    t = timed(lambda: s.solve_many_device(fk, fi), calls)
    err = float(((fi - fi_seq).abs().amax(dim=(0, 1)) / fi_seq.abs().amax(dim=(0, 1))).max())
    Bx = B_many + (16 * nk * 8 if arm == "op" else 8 * nk * dim + 8 * dim + 20) / R
    print("  %-4s: %.3f ms per field -> %.3e fits/s (%.0f B/fit -> %.0f GB/s, %.1f%% of 8 TB/s); first call %.1f ms; max col rel diff vs seq %.1e   [%s]"
          % (arm, t / R * 1e3, n * R / t, Bx, Bx * n * R / t / 1e9, Bx * n * R / t / 1e9 / 80, t_first * 1e3, err, kern))
    # oracle + 80-bit check of the last field on 512 cases
    from oracle import oracle
    import _parity
    ns = min(n, 512); r = R - 1
    xk_h = S[hoods[:ns]]; fk_h = fk[r, :ns].cpu().numpy(); xi_h = S[:ns]
    fi_in = fi0[r, :ns].cpu().numpy(); fo = fi_in.copy()
    meta = (np.full(ns, nk, np.int32), np.full(ns, order, np.int32), np.full(ns, cfg["knowns"], np.int64), np.full(ns, cfg["wm"], np.int32))
    oracle.fit_many(dim, xk_h, fk_h, meta[0], xi_h, fo, None, 0, meta[1], meta[2], meta[3])
    truth = _parity.truth_fit(dim, xk_h, fk_h, meta[0], xi_h, fi_in, meta[1], meta[2], meta[3])
    print("        vs 80-bit solution: %s %.2e, oracle %.2e, seq %.2e" % (arm, _parity.column_metric(fi[r, :ns].cpu().numpy(), truth).max(),
          _parity.column_metric(fo, truth).max(), _parity.column_metric(fi_seq[r, :ns].cpu().numpy(), truth).max()))
os.environ.pop("WLSQM_HIP_SOLVE_MANY", None)
