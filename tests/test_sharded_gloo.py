"""N > 1 path on CPU: two gloo ranks shard one cloud by point ownership, fit their blocks and
all-gather the new point values each step; the result must equal the single-process run.  The fit
itself is injected (the CPU oracle stands in for the HIP kernel, which needs a GPU): what is under
test is the partition, the index gathers and the collective."""
import os
import socket

import numpy as np
import pytest

import _cases as K
import synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_fit(dimension, order, xk, fk, nk, xi, fi, knowns, wm):
    from oracle import oracle
    n = nk.shape[0]
    fi_np = fi.numpy()
    oracle.fit_many(dimension, xk.numpy(), fk.numpy(), nk.numpy(), xi.numpy(), fi_np, None, 0,
                    np.full(n, order, np.int32), knowns.numpy(), wm.numpy())


def _run(rank, world, port, S, hoods, F0, steps, out_dir):
    import torch
    import torch.distributed as dist
    from wlsqm.sharded import ShardedCloudSolver, case_range
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    s = ShardedCloudSolver(2, S, hoods, order=2, knowns=1, weighting_method=2, device="cpu", fit_fn=_oracle_fit)
    assert (s.lo, s.hi) == case_range(len(S), rank, world)
    F = torch.from_numpy(F0.copy())
    for _ in range(steps):
        fi = s.fit(F)
        # toy explicit step: F <- F + 1e-4 * (d2F/dx2 + d2F/dy2), needs everyone's new values
        F = s.allgather_values(fi[:, 0] + 1e-4 * (fi[:, 3] + fi[:, 5]))
    np.save(os.path.join(out_dir, "F_%d_of_%d.npy" % (rank, world)), F.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_case_range_partitions():
    from wlsqm.sharded import case_range
    for n, w in ((10, 3), (1000001, 8), (5, 8), (16, 2)):
        spans = [case_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_time_stepping_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    N, nk, steps = 1501, 12, 3                                    # odd N: uneven shards exercise the padding
    S = synth.halton(N, 2)
    hoods = synth.knn(S, nk, workers=1).astype(np.int64)
    F0 = synth.field(S)
    _run(0, 1, 0, S, hoods, F0, steps, str(tmp_path))
    port = _free_port()
    mp.spawn(_run, args=(2, port, S, hoods, F0, steps, str(tmp_path)), nprocs=2, join=True)
    ref = np.load(tmp_path / "F_0_of_1.npy")
    for r in range(2):
        got = np.load(tmp_path / ("F_%d_of_2.npy" % r))
        assert np.array_equal(got, ref)                           # same arithmetic per case -> bit-identical


# ----------------------------------------------------------------------------------------------------------------------
# halo-only exchange (HaloCloudSolver): own-points-only neighbour search against a verified halo band, local tables,
# all_to_all_single of the halo values, interior / boundary split

def _cpu_knn(cand, k, nquery):
    import torch
    from scipy.spatial import cKDTree
    c = cand.numpy()
    if c.ndim == 1:
        c = c[:, None]
    _, idx = cKDTree(c).query(c[:nquery], k + 1)
    return torch.from_numpy(np.ascontiguousarray(idx[:, 1:]).astype(np.int64))


def _oracle_cloud_fit(dimension, order, S_tab, values, hoods, fi, nk, knowns, wm, pidx):
    from oracle import oracle
    n = nk.shape[0]
    S = S_tab.numpy(); V = values.numpy(); h = hoods.numpy().astype(np.int64); p = pidx.numpy().astype(np.int64)
    xk = S[h]; fk = V[h]; xi = S[p]
    fi_np = fi.numpy()                                           # a view: the oracle writes the solver's rows in place
    oracle.fit_many(dimension, xk, fk, nk.numpy(), xi, fi_np, None, 0, np.full(n, order, np.int32), knowns.numpy(), wm.numpy())


def _run_halo(rank, world, port, S, nk, F0, steps, out_dir, expect_widening=False):
    import torch
    import torch.distributed as dist
    from wlsqm.sharded import HaloCloudSolver, case_range
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dim = S.shape[1]
    s = HaloCloudSolver(dim, S, nk, order=2, knowns=1, weighting_method=2, device="cpu", fit_fn=_oracle_cloud_fit, knn_fn=_cpu_knn)
    assert (s.lo, s.hi) == case_range(len(S), rank, world)
    if world > 1:
        assert 0 < s.n_halo < len(S) - s.n_own               # a band, not the rest of the cloud
        assert 0 < s.n_int < s.n_own                         # interior and boundary cases both exist
        assert sum(s.recv_splits) == s.n_halo and s.recv_splits[rank] == 0
        attempts = [None] * world
        dist.all_gather_object(attempts, s.halo_attempts)
        if expect_widening:
            assert max(attempts) >= 2, attempts                 # some rank had to widen its band
    s.set_own_values_from_global(torch.from_numpy(F0))
    for _ in range(steps):
        fi = s.step()
        lap = (fi[:, 3] + fi[:, 5]) if dim == 2 else (fi[:, 4] + fi[:, 6] + fi[:, 8])       # i2_X2 + i2_Y2 / i3_X2 + i3_Y2 + i3_Z2
        s.values[: s.n_own] = fi[:, 0] + 1e-4 * lap                        # toy explicit step on the owned points
    g, v = s.own_values_global()
    np.save(os.path.join(out_dir, "halo_g_%d_of_%d.npy" % (rank, world)), g.numpy())
    np.save(os.path.join(out_dir, "halo_v_%d_of_%d.npy" % (rank, world)), v.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dim,clustered", [(2, 2, False), (3, 2, False), (2, 3, False), (4, 3, False), (2, 2, True), (3, 3, True)])
def test_halo_exchange_time_stepping_equals_single_process(tmp_path, world, dim, clustered):
    import torch.multiprocessing as mp
    N, nk, steps = (1501, 12, 3) if dim == 2 else (2003, 20, 2)   # odd point counts: uneven shards
    S = synth.halton(N, dim)
    if clustered:
        S = S ** 2.5              # density varies by orders of magnitude: the first halo radius (from the mean density) is too
                                  # small in the sparse corner, so the verified widening of the band has to kick in
    S = np.ascontiguousarray(S[synth.morton_order(S)])            # contiguous blocks = compact regions
    F0 = synth.field(S)
    _run_halo(0, 1, 0, S, nk, F0, steps, str(tmp_path))
    port = _free_port()
    mp.spawn(_run_halo, args=(world, port, S, nk, F0, steps, str(tmp_path), clustered), nprocs=world, join=True)
    ref = np.empty(N); ref[np.load(tmp_path / "halo_g_0_of_1.npy")] = np.load(tmp_path / "halo_v_0_of_1.npy")
    got = np.full(N, np.nan)
    for r in range(world):
        got[np.load(tmp_path / ("halo_g_%d_of_%d.npy" % (r, world)))] = np.load(tmp_path / ("halo_v_%d_of_%d.npy" % (r, world)))
    assert np.array_equal(got, ref)                               # same neighbours in the same order, same arithmetic per case


def _run_halo_blocks(rank, world, port, S, nk, F0, steps, out_dir, blocks):
    """As _run_halo, but every rank hands the solver ITS OWN block only (own_range form): no rank ever holds the whole cloud."""
    import torch
    import torch.distributed as dist
    from wlsqm.sharded import HaloCloudSolver
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lo, hi = blocks[rank]
    s = HaloCloudSolver(S.shape[1], S[lo:hi].copy(), nk, order=2, knowns=1, weighting_method=2, device="cpu", fit_fn=_oracle_cloud_fit,
                        knn_fn=_cpu_knn, own_range=(lo, len(S)))
    assert (s.lo, s.hi, s.n_own) == (lo, hi, hi - lo)
    s.set_own_values(torch.from_numpy(F0[lo:hi].copy()))
    for _ in range(steps):
        fi = s.step()
        s.values[: s.n_own] = fi[:, 0] + 1e-4 * (fi[:, 3] + fi[:, 5])
    g, v = s.own_values_global()
    np.save(os.path.join(out_dir, "blk_g_%d.npy" % rank), g.numpy())
    np.save(os.path.join(out_dir, "blk_v_%d.npy" % rank), v.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("blocks", [[(0, 700), (700, 1501)], [(0, 400), (400, 400), (400, 1501)]])
def test_halo_solver_from_own_blocks_only(tmp_path, blocks):
    """Uneven blocks chosen by the caller, one of them EMPTY: the set-up exchanges boxes, band points and need lists between the
    ranks (no rank sees the whole cloud) and the time-stepped result equals the one-process run bit for bit."""
    import torch.multiprocessing as mp
    N, nk, steps = 1501, 12, 3
    S = synth.halton(N, 2)
    S = np.ascontiguousarray(S[synth.morton_order(S)])
    F0 = synth.field(S)
    _run_halo(0, 1, 0, S, nk, F0, steps, str(tmp_path))
    world = len(blocks)
    mp.spawn(_run_halo_blocks, args=(world, _free_port(), S, nk, F0, steps, str(tmp_path), blocks), nprocs=world, join=True)
    ref = np.empty(N); ref[np.load(tmp_path / "halo_g_0_of_1.npy")] = np.load(tmp_path / "halo_v_0_of_1.npy")
    got = np.full(N, np.nan)
    for r in range(world):
        got[np.load(tmp_path / ("blk_g_%d.npy" % r))] = np.load(tmp_path / ("blk_v_%d.npy" % r))
    assert np.array_equal(got, ref)


def _run_loopback(rank, world, port, S, nk, F0):
    import torch
    import torch.distributed as dist
    from wlsqm.sharded import HaloCloudSolver
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    s = HaloCloudSolver(2, S, nk, order=2, knowns=1, weighting_method=2, device="cpu", fit_fn=_oracle_cloud_fit, knn_fn=_cpu_knn)
    s.set_own_values_from_global(torch.from_numpy(F0))
    idx = torch.arange(0, s.n_own, 7)
    s.install_loopback_halo(idx)
    s.exchange_begin(); s.exchange_end()
    assert torch.equal(s.values[s.n_own:], s.values[idx])
    dist.destroy_process_group()


def test_loopback_halo_hook_on_a_one_rank_group():
    """The hook the GPU test uses to drive the step's all_to_all_single over RCCL with one rank (tests/test_gpu_rccl.py)."""
    import torch.multiprocessing as mp
    S = synth.halton(600, 2)
    mp.spawn(_run_loopback, args=(1, _free_port(), S, 10, synth.field(S)), nprocs=1, join=True)


def test_halo_solver_on_a_lattice_with_ties(tmp_path):
    """A regular grid: dozens of candidates tie at the k-th distance of every point.  The neighbour SET must not depend on the
    partition (the cut to k happens after the canonical (distance, global index) sort): bit-identical to one rank."""
    import torch.multiprocessing as mp
    g = np.arange(40) / 39.0
    S = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    S = np.ascontiguousarray(S[synth.morton_order(S)])
    N, nk, steps = len(S), 10, 2                                  # 10 of the 12 points at the two nearest lattice distances
    F0 = synth.field(S)
    _run_halo(0, 1, 0, S, nk, F0, steps, str(tmp_path))
    mp.spawn(_run_halo, args=(3, _free_port(), S, nk, F0, steps, str(tmp_path), False), nprocs=3, join=True)
    ref = np.empty(N); ref[np.load(tmp_path / "halo_g_0_of_1.npy")] = np.load(tmp_path / "halo_v_0_of_1.npy")
    got = np.full(N, np.nan)
    for r in range(3):
        got[np.load(tmp_path / ("halo_g_%d_of_3.npy" % r))] = np.load(tmp_path / ("halo_v_%d_of_3.npy" % r))
    assert np.array_equal(got, ref)
