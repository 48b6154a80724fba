#!/usr/bin/env python3
"""BASELINE configs[4] in its literal form: ONE 3D cloud partitioned over the GPUs of a node, time-stepped, with an RCCL
all-gather of the owned point values per step (wlsqm.sharded.ShardedCloudSolver; index-based kernels, nothing dense is
materialised).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sharded.py [points_per_gpu [steps]]

(or plain `python tools/bench_sharded.py` for one GPU).  Prints one JSON line on rank 0: whole-job fits/s and the split of a
step into fit and all-gather.  bench.py --gpus N remains the driver's benchmark (independent shards, no collective)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import synth
import wlsqm.hip as whip
from wlsqm.sharded import ShardedCloudSolver

n_local = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(local_rank); dev = torch.device("cuda", local_rank)
dist = None
if "RANK" in os.environ:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
dim, order, nk = 3, 2, 40
N = n_local * world
S = synth.halton(N, dim)
S = np.ascontiguousarray(S[synth.morton_order(S)])                  # spatially compact shards
S_d = torch.from_numpy(S).to(dev)
hoods = whip.knn(S_d, nk)                                           # every rank searches the global cloud on its own GPU
solver = ShardedCloudSolver(dim, S_d, hoods.long(), order=order, knowns=1, weighting_method=2, device=dev)
del hoods
F = torch.sin(np.pi * S_d[:, 0]) * torch.cos(np.pi * S_d[:, 1]) * torch.exp(S_d[:, 2])
def sync():
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(); torch.cuda.synchronize()
def step(F):
    a = time.perf_counter(); fi = solver.fit(F); torch.cuda.synchronize(); b = time.perf_counter()
    # explicit diffusion step on the owned points, then every rank gets the new global field
    own = fi[:, 0] + 1e-7 * (fi[:, 4] + fi[:, 6] + fi[:, 8])       # i3_X2, i3_Y2, i3_Z2
    F = solver.allgather_values(own.contiguous()); torch.cuda.synchronize(); c = time.perf_counter()
    return F, b - a, c - b
for _ in range(3):                                                  # untimed: kernels, torch's elementwise ops, RCCL rings
    F, _, _ = step(F)
sync(); t_fit = t_ag = 0.0; t0 = time.perf_counter()
for _ in range(steps):
    F, df, da = step(F)
    t_fit += df; t_ag += da
sync(); dt = time.perf_counter() - t0
if dist is not None:
    t = torch.tensor([dt, t_fit, t_ag], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt, t_fit, t_ag = t.tolist()
if rank == 0:
    print(json.dumps({"metric": "local fits/s (whole node), one sharded cloud, time-stepped", "value": N * steps / dt, "unit": "fits/s",
                      "n_gpus": world, "points": N, "points_per_gpu": n_local, "steps": steps, "ms_per_step": dt / steps * 1e3,
                      "ms_fit": t_fit / steps * 1e3, "ms_allgather_and_update": t_ag / steps * 1e3,
                      "allgather_bytes_per_rank_per_step": 8 * n_local, "finite": bool(torch.isfinite(F).all())}))
if dist is not None:
    dist.barrier(); dist.destroy_process_group()
