#!/usr/bin/env python3
"""Time the basic fit of ANY shape on device-resident dense input sorted by distance (a k-nearest-neighbour search's order).
usage: python tools/time_shape.py DIM ORDER [ncases] [K] [reps] [unsorted]   -> ms per call, kernel name, fraction of the 8 TB/s HBM peak at SURVEY section 8d's bytes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import wlsqm.hip as whip
dev = torch.device("cuda", 0)
dim, order = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 25
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
no = {2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}[dim][order]
g = torch.Generator(device=dev); g.manual_seed(1)
xi = torch.rand((n, dim), dtype=torch.float64, device=dev, generator=g)
xk = xi[:, None, :] + 0.05 * (2 * torch.rand((n, K, dim), dtype=torch.float64, device=dev, generator=g) - 1)
idx = ((xk - xi[:, None, :]) ** 2).sum(-1).argsort(dim=1)
if len(sys.argv) > 6 and sys.argv[6] == "unsorted":      # every row in random order (a ball query's)
    idx = torch.rand((n, K), device=dev, generator=g).argsort(dim=1)
xk = torch.gather(xk, 1, idx[..., None].expand(-1, -1, dim)).contiguous()
fk = torch.sin(3 * xk[..., 0]) * torch.cos(2 * xk[..., 1])
fk = (fk * torch.exp(xk[..., 2]) if dim == 3 else fk).contiguous()
nk = torch.full((n,), K, dtype=torch.int32, device=dev); kn = torch.zeros(n, dtype=torch.int64, device=dev)
wm = torch.full((n,), 2, dtype=torch.int32, device=dev)
fi = torch.zeros((n, no), dtype=torch.float64, device=dev)
run = lambda: whip.fit_many_device(dim, order, xk, fk, nk, xi, fi, kn, wm)
for _ in range(3):
    run()
name = whip.last_kernel(); torch.cuda.synchronize()
for _ in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    B = 8.0 * K * (dim + 1) + 8 * dim + 8 * no + 8 + 20
    print("%dD order %d K %d n %d [%s]: %.4f ms  frac %.3f" % (dim, order, K, n, name, ms, B * n / (ms * 1e-3) / 8e12), flush=True)
