#!/usr/bin/env python3
"""Time the fit with sensitivities (do_sens) on device-resident dense input, per (dimension, order, K) shape.
usage: python tools/time_sens.py [ncases] [dim,order,K ...]
Prints ms per launch, the kernel family, the output bytes per second, and the max abs difference to the lane kernel's rows."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import wlsqm.hip as whip
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
dev = torch.device("cuda", 0)
args = sys.argv[1:]
n = int(args[0]) if args and "," not in args[0] else 200000
shapes = [tuple(int(v) for v in a.split(",")) for a in args if "," in a] or \
    [(2, 4, 50), (2, 4, 26), (2, 3, 80), (2, 2, 160), (3, 2, 160), (3, 3, 60), (3, 4, 100), (1, 2, 100)]
iterative = os.environ.get("TIME_ITERATIVE") == "1"
rng = np.random.default_rng(0)
for dim, order, K in shapes:
    no = NDOF[dim][order]
    g = torch.Generator(device=dev); g.manual_seed(1)
    xi = torch.rand((n, dim), dtype=torch.float64, device=dev, generator=g)
    xk = (xi[:, None, :] + 0.05 * (2 * torch.rand((n, K, dim), dtype=torch.float64, device=dev, generator=g) - 1)).contiguous()
    fk = torch.sin(3 * xk[..., 0]) * torch.cos(2 * xk[..., -1])
    fk = fk.contiguous()
    xi_a, xk_a = (xi[:, 0].contiguous(), xk[..., 0].contiguous()) if dim == 1 else (xi, xk)
    nk = torch.full((n,), K, dtype=torch.int32, device=dev); nk[::7] = K - 3
    kn = torch.zeros(n, dtype=torch.int64, device=dev); kn[::5] = 1
    wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
    fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = torch.sin(3 * xi[:, 0]) * torch.cos(2 * xi[:, -1])
    sens = torch.zeros((n, K, no), dtype=torch.float64, device=dev)

    def run():
        if os.environ.get("TIME_BOTH") == "1":                        # sensitivities and refinement in one call
            whip.fit_many_device(dim, order, xk_a, fk, nk, xi_a, fi, kn, wm, sens=sens, iterative=True, max_iter=10)
        elif iterative:
            whip.fit_many_device(dim, order, xk_a, fk, nk, xi_a, fi, kn, wm, iterative=True, max_iter=10)
        else:
            whip.fit_many_device(dim, order, xk_a, fk, nk, xi_a, fi, kn, wm, sens=sens)
    run(); name = whip.last_kernel(); torch.cuda.synchronize()
    reps = 5
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out = sens.clone(); fi_new = fi.clone()
    # the lane kernel on a sample of the batch
    m = min(n, 4096)
    os.environ["WLSQM_HIP_DISABLE_TILE"] = "1"; os.environ["WLSQM_HIP_DISABLE_SENS_APPLY"] = "1"
    sens2 = torch.zeros((m, K, no), dtype=torch.float64, device=dev); fi2 = torch.zeros((m, no), dtype=torch.float64, device=dev); fi2[:, 0] = fi[:m, 0]
    if iterative:
        whip.fit_many_device(dim, order, xk_a[:m], fk[:m], nk[:m], xi_a[:m], fi2, kn[:m], wm[:m], iterative=True, max_iter=10)
    else:
        whip.fit_many_device(dim, order, xk_a[:m], fk[:m], nk[:m], xi_a[:m], fi2, kn[:m], wm[:m], sens=sens2)
    name2 = whip.last_kernel()
    os.environ.pop("WLSQM_HIP_DISABLE_TILE"); os.environ.pop("WLSQM_HIP_DISABLE_SENS_APPLY")
    a, b = out[:m], sens2
    same_nan = bool((torch.isnan(a) == torch.isnan(b)).all())
    d = torch.nan_to_num(a - b, nan=0.0).abs().amax(dim=(0, 1)); sc = torch.nan_to_num(b, nan=0.0).abs().amax(dim=(0, 1)).clamp_min(1e-300)
    dfi = ((fi_new[:m] - fi2).abs().amax(0) / fi2.abs().amax(0).clamp_min(1e-300)).max()
    gbs = n * K * no * 8 / (ms * 1e-3) / 1e9
    print("dim %d order %d K %3d no %2d: %8.3f ms per %d cases  [%s]  sens out %.0f GB/s   vs %s: nan pattern %s, col rel diff %.2e, fi %.2e"
          % (dim, order, K, no, ms, n, name, 0.0 if iterative else gbs, name2, same_nan, float((d / sc).max()), float(dfi)), flush=True)
    del xk, fk, sens, out, sens2
