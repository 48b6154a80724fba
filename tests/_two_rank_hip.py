"""Worker of tests/test_gpu_round2.py::test_hip_kernels_under_world_size_2 — run under torch.distributed.run with two ranks on
the ONE GPU of the test box (process group over gloo; RCCL needs one GPU per rank).  Every rank runs the real HIP kernels:
 (1) its case-axis shard of one dense batch (wlsqm.sharded.case_range; no collective) — the concatenation must equal the
     single-process launch bit for bit;
 (2) a time-stepped partitioned cloud (ShardedCloudSolver, index-based kernel + all-gather of owned values) against the
     single-process run of the same steps.
Rank 0 writes <out>.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))

import torch
import torch.distributed as dist

import synth
import wlsqm.hip as whip
from wlsqm.sharded import ShardedCloudSolver, case_range


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # (1) dense batch, sharded by case blocks
    n, k = 20001, 32
    p = synth.cloud_problem(2, n, k)
    nk = np.full(n, k, np.int32); kn = np.zeros(n, np.int64); wm = np.full(n, 2, np.int32)
    fi0 = np.zeros((n, 6)); fi0[:, 0] = p["F"]

    def fit(lo, hi):
        fi = t(fi0[lo:hi])
        whip.fit_many_device(2, 2, t(p["xk"][lo:hi]), t(p["fk"][lo:hi]), t(nk[lo:hi]), t(p["xi"][lo:hi]), fi, t(kn[lo:hi]), t(wm[lo:hi]))
        torch.cuda.synchronize()
        return fi.cpu(), whip.last_kernel()
    lo, hi = case_range(n, rank, world)
    mine, kernel = fit(lo, hi)
    parts = [torch.zeros((case_range(n, r, world)[1] - case_range(n, r, world)[0], 6), dtype=torch.float64) for r in range(world)]
    # gloo all_gather needs equal shapes: pad to the largest shard
    pad = max(x.shape[0] for x in parts)
    buf = torch.zeros((pad, 6), dtype=torch.float64); buf[: mine.shape[0]] = mine
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    whole = torch.cat([gathered[r][: parts[r].shape[0]] for r in range(world)])
    # (2) partitioned cloud, time-stepped
    N, kc, steps = 6001, 16, 3
    S = synth.halton(N, 2)
    hoods = synth.knn(S, kc, workers=1).astype(np.int64)
    F0 = synth.field(S)

    def run(single):
        s = ShardedCloudSolver(2, t(S), hoods, order=2, knowns=1, weighting_method=2, device=dev, single=single)
        F = t(F0.copy())
        for _ in range(steps):
            fi = s.fit(F)
            F = s.allgather_values((fi[:, 0] + 1e-4 * (fi[:, 3] + fi[:, 5])).contiguous())
        torch.cuda.synchronize()
        return F.cpu().numpy()
    got = run(False)
    res = None
    if rank == 0:
        single, k1 = fit(0, n)
        ref = run(True)
        res = {"world": world, "kernel_dense": kernel, "dense_bit_identical": bool(torch.equal(whole, single)),
               "cloud_bit_identical": bool(np.array_equal(got, ref))}
        json.dump(res, open(out + ".json", "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
