#!/usr/bin/env python3
"""One-kernel 2D order-4 fit (csrc/fit_ring.hip) against the two-kernel moment path, the CPU oracle and the extended-precision
solution, over neighbour-slot counts, ragged nk, knowns masks and batch sizes that leave partial tiles / partial solve groups.
usage (GPU box): python tools/check_ring.py [K ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
import _parity as P
import wlsqm.hip as whip
from oracle import oracle

Ks = [int(a) for a in sys.argv[1:]] or [16, 24, 32, 40, 50, 64, 80, 100]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
bad = 0
for K in Ks:
    for n, ragged, kn in ((1000, False, 1), (1037, True, 0), (37, True, 0b101), (64, False, 0), (1, False, 1)):
        S = synth.halton(20000, 2, skip=1); F = synth.field(S)
        hoods = synth.knn(S, K, workers=4)[:n]
        xk = S[hoods]; fk = F[hoods]; xi = S[:n].copy()
        nk = np.full(n, K, np.int32)
        if ragged:
            nk = rng.integers(max(K - 9, 15), K + 1, n).astype(np.int32); nk[0] = K
        order = np.full(n, 4, np.int32); knowns = np.full(n, kn, np.int64); wm = np.full(n, 2, np.int32)
        if ragged:
            wm[::3] = 1
        fi0 = np.zeros((n, 15)); fi0[:, 0] = F[:n]; fi0[:, 2] = np.pi * np.sin(np.pi * S[:n, 0]) * -np.sin(np.pi * S[:n, 1])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        res = {}
        for name, env in (("ring", None), ("two-kernel", "1")):
            if env: os.environ["WLSQM_HIP_DISABLE_RING"] = env
            else: os.environ.pop("WLSQM_HIP_DISABLE_RING", None)
            fi = t(fi0)
            whip.fit_many_device(2, 4, t(xk), t(fk), t(nk), t(xi), fi, t(knowns), t(wm))
            torch.cuda.synchronize()
            res[name] = (fi.cpu().numpy(), whip.last_kernel())
        os.environ.pop("WLSQM_HIP_DISABLE_RING", None)
        fo = fi0.copy()
        oracle.fit_many(2, xk, fk, nk, xi, fo, None, 0, order, knowns, wm)
        truth = P.truth_fit(2, xk, fk, nk, xi, fi0, order, knowns, wm)
        N = P.column_metric(fo, truth).max()
        er = P.column_metric(res["ring"][0], truth).max(); e2 = P.column_metric(res["two-kernel"][0], truth).max()
        known_ok = all(np.array_equal(res["ring"][0][:, a], fi0[:, a]) for a in range(15) if (kn >> a) & 1)
        ok = known_ok and er <= 1e-10 + 8 * N and res["ring"][1] == "tile-solve"
        bad += not ok
        print("K %3d n %5d ragged %d knowns %d: kernel %-10s ring vs truth %.2e | two-kernel (%s) %.2e | oracle %.2e  %s"
              % (K, n, ragged, kn, res["ring"][1], er, res["two-kernel"][1], e2, N, "ok" if ok else "FAIL"))
print("failures:", bad)
sys.exit(1 if bad else 0)
