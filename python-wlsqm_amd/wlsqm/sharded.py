"""Case-axis sharding of one point cloud over the GPUs of a node (one process per GPU).

Local fits are independent, so a batch shards by contiguous blocks of the case axis with no
collective at all (that is what bench.py --gpus N does).  The only exchange arises when a single
global cloud is time-stepped (fk_{t+1} = F_{t+1}[hoods]): each rank owns N/world points, fits them,
and the new point values are all-gathered (8 B per point per step; RCCL over xGMI via
torch.distributed backend "nccl", gloo on CPU for the tests) — SURVEY.md §8(e).

torch is plumbing here: device memory, index gathers and the process group.
"""
__all__ = ["case_range", "ShardedCloudSolver", "HaloCloudSolver"]


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


TIE_PAD = 8          # extra candidates per point fetched by HaloCloudSolver so that ties at the k-th distance are cut canonically


def _default_knn(cand, k, nquery):
    from . import hip
    return hip.knn(cand, k, nquery=nquery).long()


class HaloCloudSolver:
    """ONE global cloud partitioned over the ranks, halo exchange only (the literal form of BASELINE configs[4]).

    Points are owned in contiguous index blocks, rank by rank (order the cloud along a space-filling curve first —
    synth.morton_order — so that a block is a compact region).  A rank holds ITS OWN block only; nothing in the set-up or in a
    step scans or stores the coordinates of the whole cloud.  Every rank
      * publishes the bounding box of its block inflated by a halo radius (all_gather of 2 dim doubles) and receives from every
        other rank exactly the points of THAT rank's block inside the box (all_to_all of coordinates + global indices); the radius
        is verified afterwards — the largest k-th neighbour distance over all ranks must not exceed it, otherwise the bands widen
        and the exchange repeats;
      * searches the neighbours of its own points only, against [own | received band] — `wlsqm.hip.knn(..., nquery=n_own)`;
      * keeps a LOCAL point table [interior own | boundary own | halo] (coordinates and values) and local neighbour lists, so the
        index-based fit kernel gathers from a table of its shard's size;
      * tells each owner which of its points it names (all_to_all of the need lists, set-up only) and per step exchanges only
        those values: one `all_to_all_single` (uneven splits; RCCL over xGMI, gloo in the CPU tests) on a side stream while the
        INTERIOR cases (all neighbours owned) are fitted; the boundary cases follow when the halo has arrived.
    Neighbour lists are put in the canonical order (distance, GLOBAL index) and cut to k AFTER that sort (the search fetches
    k + TIE_PAD candidates), so a partitioned run reproduces the one-rank run bit for bit — also on lattice-like clouds, where many
    candidates tie at the k-th distance — as long as a run of equal distances does not reach past the padded list
    (`ties_beyond_pad` counts the points where it does).

    S: either the whole cloud (N, dim), the same on every rank — the constructor keeps rows case_range(N, rank, world) and drops
    the rest (legacy form, tests) — or, with own_range=(lo, N), this rank's block only, whose first point is global point lo."""

    def __init__(self, dimension, S, nk, order, knowns, weighting_method, device, group=None, fit_fn=None, knn_fn=None,
                 single=False, own_range=None):
        import numpy as np
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group) if dist.is_initialized() and not single else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() and not single else 1
        world, rank = self.world, self.rank
        self.dimension, self.order, self.nk = int(dimension), int(order), int(nk)
        self.fit_fn = fit_fn
        knn_fn = knn_fn or _default_knn
        dev = torch.device(device)
        self.device = dev
        # gloo has no device-memory collectives: stage through the host (tests with several ranks on one GPU; RCCL takes the
        # device tensors)
        self._host_stage = bool(world > 1 and dev.type == "cuda" and dist.get_backend(group) == "gloo")
        S = torch.as_tensor(S, dtype=torch.float64)
        if S.dim() == 1:
            S = S[:, None]
        dim = int(S.shape[1])
        if own_range is None:
            N = int(S.shape[0])
            lo, hi = case_range(N, rank, world)
            own = S[lo:hi].to(dev).contiguous()
        else:
            lo, N = int(own_range[0]), int(own_range[1])
            own = S.to(dev).contiguous()
            hi = lo + int(own.shape[0])
        del S
        self.N, self.lo, self.hi = N, lo, hi
        n_own = hi - lo
        # block bounds of every rank (blocks are contiguous and ascend with the rank)
        if world > 1:
            b = self._allgather(torch.tensor([lo, hi], dtype=torch.int64, device=dev)).cpu().numpy().reshape(world, 2)
            if b[0, 0] != 0 or b[-1, 1] != N or any(b[q, 1] != b[q + 1, 0] for q in range(world - 1)):
                raise ValueError("the ranks' blocks must tile [0, N) in rank order; got %s" % b.tolist())
            bounds = np.concatenate([b[:, 0], [N]])
        else:
            bounds = np.array([0, N])
        self._bounds = bounds
        own_g = torch.arange(lo, hi, device=dev)
        # ---- 1. candidates: own points + the band received from the other ranks, verified
        if world == 1:
            cand, cand_g = own, own_g
            kq = max(self.nk, min(self.nk + TIE_PAD, N - 1))
            hl = knn_fn(own if dim > 1 else own[:, 0].contiguous(), kq, n_own)
            self.halo_radius, self.halo_attempts = 0.0, 0
        else:
            inf = float("inf")
            blo = own.min(0).values if n_own else torch.full((dim,), inf, dtype=torch.float64, device=dev)
            bhi = own.max(0).values if n_own else torch.full((dim,), -inf, dtype=torch.float64, device=dev)
            glo = self._allreduce(blo.clone(), "min"); ghi = self._allreduce(bhi.clone(), "max")
            ext = (ghi - glo).clamp_min(1e-300)
            ball = {1: 2.0, 2: np.pi, 3: 4.0 * np.pi / 3.0}[dim]
            r = 1.5 * float((self.nk * float(ext.prod()) / (N * ball)) ** (1.0 / dim))
            for attempt in range(8):
                # Who needs which of my points: round 3 sent every point inside the asking rank's BOUNDING BOX inflated by r — for a Morton
                # block 2-4x the block's own volume.  Round 5: an occupancy grid over the global extent with cells of side >= r; a rank
                # marks the cells that hold its points, grows the marks by one cell in every direction (every point within r of one of
                # its points lies in a marked cell) and publishes the marks (all_gather of one byte per cell, <= 64^dim cells); a point
                # is sent to the ranks whose marks cover its cell.  The band shrinks to the block's own outline; the search result —
                # verified against r below as before — cannot change (tests/test_sharded_gloo.py: bit-identical to one rank).
                # (cells per axis, side ext / ncell >= r (1 + 1e-9): the margin keeps a neighbour EXACTLY r away along an axis — ext / r an exact
                # integer — within one cell of its point after the floating-point floor below: ADVICE r5.  A zero extent along an axis gives
                # one cell there.)
                ncell = torch.clamp((ext / (r * (1.0 + 1e-9))).floor(), 1, {1: 4096, 2: 256, 3: 64}[dim]).to(torch.int64)
                hcell = ext / ncell.to(torch.float64)
                stride = torch.ones(dim, dtype=torch.int64, device=dev)
                for m in range(dim - 2, -1, -1):
                    stride[m] = stride[m + 1] * ncell[m + 1]
                ntot = int(ncell.prod().item())
                shape = tuple(int(v) for v in ncell.tolist())
                if n_own:
                    cell = torch.minimum(((own - glo) / hcell).floor().to(torch.int64).clamp_min(0), ncell - 1)
                    lin = (cell * stride).sum(1)
                    occ = torch.zeros(ntot, dtype=torch.float32, device=dev)
                    occ[lin] = 1.0
                    pool = {1: torch.nn.functional.max_pool1d, 2: torch.nn.functional.max_pool2d, 3: torch.nn.functional.max_pool3d}[dim]
                    occ = pool(occ.reshape((1, 1) + shape), kernel_size=3, stride=1, padding=1).reshape(-1)
                else:
                    lin = torch.zeros(0, dtype=torch.int64, device=dev)
                    occ = torch.zeros(ntot, dtype=torch.float32, device=dev)
                marks = self._allgather(occ.to(torch.uint8)).reshape(world, ntot)
                parts, counts = [], []
                for q in range(world):
                    if q == rank or n_own == 0:
                        parts.append(torch.zeros(0, dtype=torch.int64, device=dev)); counts.append(0)
                        continue
                    idx = torch.nonzero(marks[q][lin])[:, 0]
                    parts.append(idx); counts.append(int(idx.numel()))
                sel = torch.cat(parts)
                payload = torch.cat([own[sel], (own_g[sel]).to(torch.float64)[:, None]], 1).contiguous()      # indices < 2^53: exact
                got_counts = self._alltoall_counts(counts)
                recv = self._alltoall(payload.reshape(-1), [c * (dim + 1) for c in counts],
                                      [c * (dim + 1) for c in got_counts]).reshape(-1, dim + 1)
                cand = torch.cat([own, recv[:, :dim]]).contiguous()
                cand_g = torch.cat([own_g, recv[:, dim].to(torch.int64)])
                if n_own:
                    kq = max(self.nk, min(self.nk + TIE_PAD, int(cand.shape[0]) - 1))
                    hl = knn_fn(cand if dim > 1 else cand[:, 0].contiguous(), kq, n_own)
                    dk = (cand[hl[:, self.nk - 1]] - own).pow(2).sum(1).max().sqrt().reshape(1)
                else:
                    hl = torch.zeros((0, self.nk), dtype=torch.int64, device=dev)
                    dk = torch.zeros(1, dtype=torch.float64, device=dev)
                dk = float(self._allreduce(dk.to(torch.float64), "max").item())
                self.halo_attempts = attempt + 1
                if dk <= r:
                    break
                r = 1.25 * dk
            else:
                raise RuntimeError("halo band did not converge")
            self.halo_radius = r
            self.band_points_received = int(recv.shape[0])
        # ---- 2. canonical neighbour order: ascending (distance, GLOBAL index) — independent of the candidate numbering.  The search
        # returned nk + TIE_PAD candidates per point: the cut to nk happens HERE, after the canonical sort, so that a tie at the
        # nk-th distance (lattice-like clouds) is cut the same way whatever the partition; ties_beyond_pad counts the
        # points whose run of equal distances reaches past the padded list (there the set may still depend on the partition)
        hg = cand_g[hl]                                           # (n_own, kq) global indices
        d2 = (cand[hl] - own[:, None, :]).pow(2).sum(2)
        o1 = torch.argsort(hg, dim=1, stable=True)
        hg = torch.gather(hg, 1, o1); d2 = torch.gather(d2, 1, o1)
        o2 = torch.argsort(d2, dim=1, stable=True)
        hg = torch.gather(hg, 1, o2); d2 = torch.gather(d2, 1, o2)
        self.ties_beyond_pad = int((d2[:, -1] == d2[:, self.nk - 1]).sum().item()) if (n_own and hg.shape[1] > self.nk) else 0
        hg = hg[:, : self.nk].contiguous()
        del d2, o1, o2, hl
        # ---- 3. local numbering [interior own | boundary own | halo]
        foreign = (hg < lo) | (hg >= hi)
        is_bnd = foreign.any(1)
        order_own = torch.cat([torch.nonzero(~is_bnd)[:, 0], torch.nonzero(is_bnd)[:, 0]])       # local -> own offset
        self.n_own, self.n_int = n_own, int((~is_bnd).sum().item())
        need_g = torch.unique(hg[foreign])                        # sorted global indices of the halo points actually named
        self.n_halo = int(need_g.numel())
        self.gidx_own = (order_own + lo)                          # global index of local own point i
        loc_of_own = torch.empty(n_own, dtype=torch.int64, device=dev)
        loc_of_own[order_own] = torch.arange(n_own, device=dev)
        hl = torch.where(foreign, torch.zeros_like(hg), loc_of_own[(hg - lo).clamp(0, max(n_own - 1, 0))]) if n_own else hg
        if self.n_halo:
            hl = torch.where(foreign, n_own + torch.searchsorted(need_g, hg.clamp(0, N - 1)), hl)
        hl = hl[order_own]                                        # rows in local case order
        S_own = own[order_own]
        if self.n_halo:
            # coordinates of the named halo points, from the received band (every global index arrives at most once)
            rg = cand_g[n_own:]
            ro = torch.argsort(rg)
            pos = ro[torch.searchsorted(rg[ro], need_g)]
            S_tab = torch.cat([S_own, cand[n_own:][pos]])
        else:
            S_tab = S_own
        del cand, cand_g
        self.S_tab = (S_tab[:, 0] if self.dimension == 1 else S_tab).contiguous()
        self.hoods32 = hl.to(torch.int32).contiguous()
        self.values = torch.zeros(n_own + self.n_halo, dtype=torch.float64, device=dev)     # field on the local table
        no = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}[self.dimension][self.order]
        self.fi = torch.zeros((n_own, no), dtype=torch.float64, device=dev)
        self.nk_t = torch.full((n_own,), self.nk, dtype=torch.int32, device=dev)
        self.kn_t = torch.full((n_own,), int(knowns), dtype=torch.int64, device=dev)
        self.wm_t = torch.full((n_own,), int(weighting_method), dtype=torch.int32, device=dev)
        self.pidx = torch.arange(n_own, dtype=torch.int32, device=dev)
        # ---- 4. exchange plan (set-up only): every owner learns which of its points the others name.  need_g is sorted, so
        # it is already grouped by owner in rank order, and the owner answers in the order it was asked
        self.send_idx = torch.zeros(0, dtype=torch.int64, device=dev)
        self.send_splits, self.recv_splits = [0] * world, [0] * world
        if world > 1:
            mine = need_g.cpu().numpy()
            self.recv_splits = [int(np.searchsorted(mine, bounds[q + 1]) - np.searchsorted(mine, bounds[q])) for q in range(world)]
            self.send_splits = self._alltoall_counts(self.recv_splits)
            want = self._alltoall(need_g, self.recv_splits, self.send_splits)      # global indices of MY points, by asking rank
            self.send_idx = loc_of_own[want - lo]
        self._exchange_on = world > 1
        self._send = torch.empty(int(self.send_idx.numel()), dtype=torch.float64, device=dev)
        self._recv = torch.empty(self.n_halo, dtype=torch.float64, device=dev)
        self._comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    # -- collectives of the set-up (host-staged when the backend has no device collectives) -----------------------------------
    def _stage(self, t):
        return t.cpu() if self._host_stage else t

    def _allgather(self, t):
        torch, dist = self.torch, self.dist
        src = self._stage(t.contiguous())
        out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out.reshape(-1), src.reshape(-1), group=self.group)
        return out.to(self.device)

    def _allreduce(self, t, op):
        dist = self.dist
        src = self._stage(t.contiguous())
        dist.all_reduce(src, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op], group=self.group)
        return src.to(self.device)

    def _alltoall_counts(self, counts):
        torch, dist = self.torch, self.dist
        dev = "cpu" if self._host_stage or self.device.type == "cpu" else self.device
        src = torch.tensor(counts, dtype=torch.int64, device=dev)
        out = torch.empty_like(src)
        dist.all_to_all_single(out, src, group=self.group)
        return [int(v) for v in out.cpu().tolist()]

    def _alltoall(self, t, send_splits, recv_splits):
        torch, dist = self.torch, self.dist
        src = self._stage(t.contiguous())
        out = torch.empty(int(sum(recv_splits)), dtype=src.dtype, device=src.device)
        dist.all_to_all_single(out, src, list(recv_splits), list(send_splits), group=self.group)
        return out.to(self.device)

    def install_loopback_halo(self, own_local_idx):
        """Test hook: make this rank its own neighbour — the values of the local own points `own_local_idx` are sent through the
        step's all_to_all_single to THIS rank and land in freshly appended halo slots of the value table.  Lets a one-rank
        process group (e.g. RCCL on a one-GPU box) drive exchange_begin / exchange_end exactly as a partitioned run does."""
        torch = self.torch
        idx = torch.as_tensor(own_local_idx, dtype=torch.int64, device=self.device)
        m = int(idx.numel())
        self.send_idx = idx
        self.send_splits = [0] * self.world; self.recv_splits = [0] * self.world
        self.send_splits[self.rank] = m; self.recv_splits[self.rank] = m
        self.n_halo = m
        self.values = torch.cat([self.values[: self.n_own], torch.zeros(m, dtype=torch.float64, device=self.device)])
        self._send = torch.empty(m, dtype=torch.float64, device=self.device)
        self._recv = torch.empty(m, dtype=torch.float64, device=self.device)
        self._exchange_on = True

    # -- values ------------------------------------------------------------------------------------------------------------
    def set_own_values_from_global(self, F_global):
        """Own part of the local value table from a global (N,) vector."""
        self.values[: self.n_own] = F_global.to(self.device)[self.gidx_own]

    def set_own_values(self, v_block):
        """Own part of the local value table from this rank's block of the field (block order: global points lo .. hi - 1)."""
        self.values[: self.n_own] = v_block.to(self.device)[self.gidx_own - self.lo]

    def own_values_global(self):
        """(global indices, values) of the owned points."""
        return self.gidx_own, self.values[: self.n_own]

    # -- one step ----------------------------------------------------------------------------------------------------------
    def exchange_begin(self):
        """Pack the owned values the other ranks name and start the halo exchange (side stream on a GPU)."""
        torch, dist = self.torch, self.dist
        if not self._exchange_on:
            return
        torch.index_select(self.values, 0, self.send_idx, out=self._send)
        if self._host_stage:
            send = self._send.cpu(); recv = torch.empty(self.n_halo, dtype=torch.float64)
            dist.all_to_all_single(recv, send, self.recv_splits, self.send_splits, group=self.group)
            self.values[self.n_own:] = recv.to(self.device)
            self._halo_ready = torch.cuda.Event(); self._halo_ready.record()
            return
        if self._comm_stream is not None:
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                dist.all_to_all_single(self._recv, self._send, self.recv_splits, self.send_splits, group=self.group)
                self.values[self.n_own:] = self._recv
                self._halo_ready = torch.cuda.Event(); self._halo_ready.record()
        else:
            dist.all_to_all_single(self._recv, self._send, self.recv_splits, self.send_splits, group=self.group)
            self.values[self.n_own:] = self._recv

    def exchange_end(self):
        if self._exchange_on and self._comm_stream is not None:
            self.torch.cuda.current_stream().wait_event(self._halo_ready)

    def _fit_rows(self, a, b):
        if b <= a:
            return
        self.fi[a:b, 0] = self.values[a:b]
        if self.fit_fn is None:
            from . import hip
            hip.fit_cloud_device(self.dimension, self.order, self.S_tab, self.values, self.hoods32[a:b], self.fi[a:b],
                                 self.nk_t[a:b], self.kn_t[a:b], self.wm_t[a:b], point_index=self.pidx[a:b])
        else:
            self.fit_fn(self.dimension, self.order, self.S_tab, self.values, self.hoods32[a:b], self.fi[a:b], self.nk_t[a:b],
                        self.kn_t[a:b], self.wm_t[a:b], self.pidx[a:b])

    def fit_interior(self):
        self._fit_rows(0, self.n_int)

    def fit_boundary(self):
        self._fit_rows(self.n_int, self.n_own)

    def step(self):
        """Exchange the halo and fit every owned point; the interior fits overlap the exchange.  Returns fi (n_own, no) in
        LOCAL order (gidx_own names the global point of each row)."""
        self.exchange_begin()
        self.fit_interior()
        self.exchange_end()
        self.fit_boundary()
        return self.fi
