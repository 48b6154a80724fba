"""Case-axis sharding of one point cloud over the GPUs of a node (one process per GPU).

Local fits are independent, so a batch shards by contiguous blocks of the case axis with no
collective at all (that is what bench.py --gpus N does).  The only exchange arises when a single
global cloud is time-stepped (fk_{t+1} = F_{t+1}[hoods]): each rank owns N/world points, fits them,
and the new point values are all-gathered (8 B per point per step; RCCL over xGMI via
torch.distributed backend "nccl", gloo on CPU for the tests) — SURVEY.md §8(e).

torch is plumbing here: device memory, index gathers and the process group.
"""
__all__ = ["case_range", "ShardedCloudSolver"]


def case_range(ncases, rank, world):
    """Contiguous block [lo, hi) of the case axis owned by `rank`; the first ncases % world ranks get one extra."""
    base, rem = divmod(int(ncases), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)



class ShardedCloudSolver:
    """Fits every point of ONE global cloud, sharded by point ownership.

    S (N, dim) float64 and hoods (N, nk) int64 are the global coordinates and neighbour lists (each rank
    passes the same arrays; only its own rows of `hoods` are kept).  order/knowns/weighting are uniform.
    """

    def __init__(self, dimension, S, hoods, order, knowns, weighting_method, device, group=None, fit_fn=None, single=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        # single=True: own the whole cloud even inside an initialised process group (the one-process answer a sharded run
        # is compared with)
        self.rank = dist.get_rank(group) if dist.is_initialized() and not single else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() and not single else 1
        self.dimension, self.order = int(dimension), int(order)
        self.N, self.nk = int(hoods.shape[0]), int(hoods.shape[1])
        self.lo, self.hi = case_range(self.N, self.rank, self.world)
        self.fit_fn = fit_fn          # None: index-based HIP kernel (no dense xk/fk is ever materialised)
        n = self.hi - self.lo
        S = torch.as_tensor(S, dtype=torch.float64)
        if S.dim() == 1:
            S = S[:, None]
        self.S = S.to(device)
        self.hoods = torch.as_tensor(hoods[self.lo:self.hi], dtype=torch.int64).to(device)
        if fit_fn is None:
            self.hoods32 = self.hoods.to(torch.int32).contiguous()
            self.pidx = torch.arange(self.lo, self.hi, dtype=torch.int32, device=device)
            self.S_tab = (self.S[:, 0] if self.dimension == 1 else self.S).contiguous()
        else:
            # injected (dense) fit: the geometry of the owned cases is gathered once
            xk = self.S[self.hoods]                               # (n, nk, dim)
            self.xk = (xk[..., 0] if self.dimension == 1 else xk).contiguous()
            xi = self.S[self.lo:self.hi]
            self.xi = (xi[:, 0] if self.dimension == 1 else xi).contiguous()
        self.nk_t = torch.full((n,), self.nk, dtype=torch.int32, device=device)
        self.kn_t = torch.full((n,), int(knowns), dtype=torch.int64, device=device)
        self.wm_t = torch.full((n,), int(weighting_method), dtype=torch.int32, device=device)
        no = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}[self.dimension][self.order]
        self.fi = torch.zeros((n, no), dtype=torch.float64, device=device)
        self._pad = max(case_range(self.N, r, self.world)[1] - case_range(self.N, r, self.world)[0]
                        for r in range(self.world))

    def fit(self, F_global):
        """One pass: fit all owned points to the global point values F_global (N,).  Returns fi (n_own, no)."""
        self.fi[:, 0] = F_global[self.lo:self.hi]
        if self.fit_fn is None:
            from . import hip
            hip.fit_cloud_device(self.dimension, self.order, self.S_tab, F_global.contiguous(), self.hoods32, self.fi,
                                 self.nk_t, self.kn_t, self.wm_t, point_index=self.pidx)
        else:
            fk = F_global[self.hoods].contiguous()
            self.fit_fn(self.dimension, self.order, self.xk, fk, self.nk_t, self.xi, self.fi, self.kn_t, self.wm_t)
        return self.fi

    def allgather_values(self, v_own):
        """All-gather one float64 per owned point into the global (N,) vector on every rank."""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            return v_own.clone()
        buf = torch.zeros(self._pad, dtype=v_own.dtype, device=v_own.device)
        buf[: v_own.shape[0]] = v_own
        out = torch.empty(self._pad * self.world, dtype=v_own.dtype, device=v_own.device)
        dist.all_gather_into_tensor(out, buf, group=self.group)
        parts = []
        for r in range(self.world):
            lo, hi = case_range(self.N, r, self.world)
            parts.append(out[r * self._pad: r * self._pad + (hi - lo)])
        return torch.cat(parts)
