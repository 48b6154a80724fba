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
from wlsqm.sharded import HaloCloudSolver, ShardedCloudSolver, case_range


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

    # (3) the same with the halo-only exchange: own-points-only GPU neighbour search against a halo band, local tables,
    # all_to_all of the halo values, interior / boundary launches (wlsqm.sharded.HaloCloudSolver) -- 3D, 40 neighbours
    N3, k3 = 30011, 40
    S3 = synth.halton(N3, 3)
    S3 = np.ascontiguousarray(S3[synth.morton_order(S3)])
    F3 = synth.field(S3)

    def run_halo(single):
        s = HaloCloudSolver(3, t(S3), k3, order=2, knowns=0, weighting_method=2, device=dev, single=single)
        s.set_own_values_from_global(t(F3))
        for _ in range(steps):
            fi = s.step()
            s.values[: s.n_own] = fi[:, 0] + 1e-7 * (fi[:, 4] + fi[:, 6] + fi[:, 8])
        torch.cuda.synchronize()
        g, v = s.own_values_global()
        return g.cpu().numpy(), v.cpu().numpy(), whip.last_kernel(), (s.n_halo, s.n_int, s.n_own)
    g_mine, v_mine, k_halo, shape = run_halo(False)
    sizes = [None] * world
    dist.all_gather_object(sizes, (g_mine, v_mine))
    res = None
    if rank == 0:
        single, k1 = fit(0, n)
        ref = run(True)
        g1, v1, _, _ = run_halo(True)
        ref3 = np.empty(N3); ref3[g1] = v1
        got3 = np.full(N3, np.nan)
        for g_r, v_r in sizes:
            got3[g_r] = v_r
        res = {"world": world, "kernel_dense": kernel, "dense_bit_identical": bool(torch.equal(whole, single)),
               "cloud_bit_identical": bool(np.array_equal(got, ref)), "halo_bit_identical": bool(np.array_equal(got3, ref3)),
               "halo_kernel": k_halo, "halo_shape_rank0": list(shape)}
        json.dump(res, open(out + ".json", "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
