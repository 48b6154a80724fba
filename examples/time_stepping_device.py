#!/usr/bin/env python3
"""Everything on the GPU: neighbour search, geometry prepared once, one fused launch per time step, stacked fields in one launch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-wlsqm_amd"))
import wlsqm
import wlsqm.hip

n, nk, order = 200_000, 32, 2
rng = np.random.default_rng(1)
S = rng.uniform(0.0, 1.0, (n, 2))
dev = torch.device("cuda", 0)
S_d = torch.from_numpy(S).to(dev)
h_d = wlsqm.hip.knn(S_d, nk).long()                                  # nk nearest neighbours of every point, on the GPU
solver = wlsqm.ExpertSolver(dimension=2, nk=np.full(n, nk, np.int32), order=np.full(n, order, np.int32),
                            knowns=np.full(n, wlsqm.b2_F, np.int64),
                            weighting_method=np.full(n, wlsqm.WEIGHT_CENTER, np.int32))
solver.prepare_device(S_d, S_d[h_d].contiguous())                   # geometry device to device: nothing crosses PCIe

no = wlsqm.number_of_dofs(2, order)
fi = torch.zeros((n, no), dtype=torch.float64, device=dev)

# a travelling wave u(x, y, t) = sin(pi (x - t)) cos(pi y): every step gathers the new values on the device, refits all
# local models on the resident geometry and checks the advection residual u_t + u_x = 0 with the fitted derivative
inner = ((S_d - 0.5).abs() < 0.45).all(dim=1)
steps, dt, worst = 200, 1e-3, 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
for step in range(steps):
    t = step * dt
    u = torch.sin(np.pi * (S_d[:, 0] - t)) * torch.cos(np.pi * S_d[:, 1])
    fi[:, 0] = u
    solver.solve_device(u[h_d], fi)                              # fk = u[hoods] gathered on the device
    u_t = -np.pi * torch.cos(np.pi * (S_d[:, 0] - t)) * torch.cos(np.pi * S_d[:, 1])
    worst = max(worst, float((u_t + fi[:, wlsqm.i2_X])[inner].abs().max()))
torch.cuda.synchronize(); t1 = time.perf_counter()
print("%d time steps on %d points: %.2f ms per step (gather + fit + check), max advection residual %.2e"
      % (steps, n, (t1 - t0) / steps * 1e3, worst))

# many independent fields on the same geometry in one launch
R = 16
fk = torch.stack([(torch.sin(np.pi * S_d[:, 0] + 0.1 * r) * torch.cos(np.pi * S_d[:, 1]))[h_d] for r in range(R)])
fis = torch.zeros((R, n, no), dtype=torch.float64, device=dev)
fis[:, :, 0] = torch.stack([torch.sin(np.pi * S_d[:, 0] + 0.1 * r) * torch.cos(np.pi * S_d[:, 1]) for r in range(R)])
solver.solve_many_device(fk, fis)
torch.cuda.synchronize()
err = max(float((fis[r, inner, wlsqm.i2_X] - np.pi * torch.cos(np.pi * S_d[inner, 0] + 0.1 * r) * torch.cos(np.pi * S_d[inner, 1])).abs().max())
          for r in range(R))
print("solve_many_device, %d fields: max |df/dx error| = %.2e" % (R, err))

# an explicit time integration (u_t = -u_x, forward Euler) with the whole step captured into a HIP graph: the device entry
# points only enqueue work on the current stream, so gather + fit + update replay as one graph launch per step
def advect(u, fk_buf, dt_):
    fk_buf.copy_(u[h_d])
    fi[:, 0] = u
    solver.solve_device(fk_buf, fi)
    u.sub_(dt_ * fi[:, wlsqm.i2_X])

u0 = torch.sin(np.pi * S_d[:, 0]) * torch.cos(np.pi * S_d[:, 1])
fk_buf = torch.empty((n, nk), dtype=torch.float64, device=dev)
steps, dt = 200, 2e-5
u = u0.clone()
advect(u, fk_buf, dt)                                               # warm-up outside the capture
u.copy_(u0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    advect(u, fk_buf, dt)
torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / steps
u_eager = u.clone()
u.copy_(u0)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
    advect(u, fk_buf, dt)
u.copy_(u0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    graph.replay()
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / steps
exact = torch.sin(np.pi * (S_d[:, 0] - steps * dt)) * torch.cos(np.pi * S_d[:, 1])
print("forward-Euler advection, %d steps: eager %.3f ms per step, one HIP graph replay per step %.3f ms; identical: %s; "
      "max error against the exact solution %.2e"
      % (steps, t_eager * 1e3, t_graph * 1e3, bool(torch.equal(u, u_eager)), float((u - exact)[inner].abs().max())))
