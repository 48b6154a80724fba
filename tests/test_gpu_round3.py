"""GPU tests added in round 3 (beside test_gpu_strict.py and test_gpu_rccl.py): the repack / gather scratch in bounded slices
(ADVICE r2), the logical neighbour count behind a repack, per-case order tensors on the device-resident API."""
import os

import numpy as np
import pytest

import _cases as K
import _parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wlsqm():
    import wlsqm as W
    from wlsqm import _binding
    assert _binding.lib().wlsqm_hip_device_count() >= 1, "no HIP device: the GPU tests need a real MI355X"
    return W


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _t(a, dev="cuda:0"):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _batch(dim, order, Kn, n, seed):
    rng = np.random.default_rng(seed)
    no = K.NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim))
    xk = xi[:, None, :] + 0.08 * rng.uniform(-1, 1, (n, Kn, dim))
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
    nk = rng.integers(max(no + 2, Kn - 6), Kn + 1, n).astype(np.int32); nk[0] = Kn
    fi0 = np.zeros((n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
    return dict(xi=xi, xk=xk, fk=fk, nk=nk, kn=np.zeros(n, np.int64), wm=np.full(n, 2, np.int32), fi0=fi0, no=no)


@pytest.mark.parametrize("dim,order,Kn,sens", [(2, 2, 31, False), (3, 2, 39, False), (2, 2, 33, True), (2, 4, 51, False)])
def test_repack_in_bounded_slices_equals_one_pass(wlsqm, dim, order, Kn, sens, monkeypatch):
    """Odd-K dense rows go through the repack scratch; with WLSQM_HIP_REPACK_MB=1 the batch needs several slices of the SAME
    scratch block: bit-identical to the one-slice run (the slices are independent cases), fi and sens."""
    import torch
    import wlsqm.hip as whip
    n = 9000
    b = _batch(dim, order, Kn, n, Kn)
    args = lambda fi: (dim, order, _t(b["xk"]), _t(b["fk"]), _t(b["nk"]), _t(b["xi"]), fi, _t(b["kn"]), _t(b["wm"]))
    out = {}
    for mb in ("512", "1"):
        monkeypatch.setenv("WLSQM_HIP_REPACK_MB", mb)
        fi = _t(b["fi0"])
        s = torch.full((n, Kn, b["no"]), 7.0, dtype=torch.float64, device="cuda:0") if sens else None
        whip.fit_many_device(*args(fi), sens=s)
        torch.cuda.synchronize()
        assert whip.last_kernel() != "lane", whip.last_kernel()
        out[mb] = (fi.cpu().numpy(), None if s is None else s.cpu().numpy())
    assert np.array_equal(out["1"][0], out["512"][0])
    if sens:
        assert np.array_equal(out["1"][1], out["512"][1], equal_nan=True)
    truth = P.truth_fit(dim, b["xk"], b["fk"], b["nk"], b["xi"], b["fi0"], np.full(n, order, np.int32), b["kn"], b["wm"])
    fi_o = b["fi0"].copy()
    from oracle import oracle
    oracle.fit_many(dim, b["xk"], b["fk"], b["nk"], b["xi"], fi_o, None, 0, np.full(n, order, np.int32), b["kn"], b["wm"], ntasks=8)
    P.assert_parity(out["1"][0], fi_o, truth, "sliced repack vs oracle")


@pytest.mark.parametrize("with_pidx", [False, True])
def test_gather_in_bounded_slices(wlsqm, with_pidx, monkeypatch):
    """Index-based input of a shape without a gather kernel (2D order 4 with an odd K is gathered into dense scratch): slices of
    the scratch must keep every case's OWN point as xi — with and without point_index."""
    import torch
    import synth
    import wlsqm.hip as whip
    N, Kn = 9000, 51
    S = synth.halton(N, 2)
    S = np.ascontiguousarray(S[synth.morton_order(S)])
    F = synth.field(S)
    hoods = synth.knn(S, Kn, workers=4).astype(np.int32)
    n = N
    nk = np.full(n, Kn, np.int32); kn = np.ones(n, np.int64); wm = np.full(n, 2, np.int32)
    fi0 = np.zeros((n, 15)); fi0[:, 0] = F
    pidx = _t(np.arange(n, dtype=np.int32)) if with_pidx else None
    out = {}
    for mb in ("512", "1"):
        monkeypatch.setenv("WLSQM_HIP_REPACK_MB", mb)
        fi = _t(fi0)
        whip.fit_cloud_device(2, 4, _t(S), _t(F), _t(hoods), fi, _t(nk), _t(kn), _t(wm), point_index=pidx)
        torch.cuda.synchronize()
        out[mb] = fi.cpu().numpy()
    assert np.array_equal(out["1"], out["512"])
    fi_o = fi0.copy()
    from oracle import oracle
    h = hoods.astype(np.int64)
    oracle.fit_many(2, S[h], F[h], nk, S, fi_o, None, 0, np.full(n, 4, np.int32), kn, wm, ntasks=8)
    truth = P.truth_fit(2, S[h][:512], F[h][:512], nk[:512], S[:512], fi0[:512], np.full(512, 4, np.int32), kn[:512], wm[:512])
    P.assert_parity(out["1"][:512], fi_o[:512], truth, "sliced gather vs oracle")
    # cases of the LAST slice too (their xi must be their own point, not the slice-local row number)
    E = P.column_metric(out["1"][-512:], fi_o[-512:])
    N_ = P.column_metric(fi_o[:512], truth)
    assert np.all(E <= 1e-10 + 64 * N_), (E, N_)


def test_bad_nk_behind_a_repack_is_clamped_to_the_callers_k(wlsqm):
    """nk[j] > K is invalid input the header promises to clamp to the extent of the neighbour axis.  Behind a repack the slot
    count is K + 1 for odd K: the clamp must still be K (the pad slot is not a neighbour), and with do_sens no row K may be
    written — for the last case that row would be out of bounds."""
    import torch
    import wlsqm.hip as whip
    dim, order, Kn, n = 2, 2, 31, 600
    b = _batch(dim, order, Kn, n, 5)
    nk_bad = b["nk"].copy(); nk_bad[::3] = Kn + 40
    nk_ok = np.minimum(nk_bad, Kn).astype(np.int32)
    res = {}
    for name, nk in (("bad", nk_bad), ("ok", nk_ok)):
        fi = _t(b["fi0"])
        guard = torch.full((n + 1, Kn, 6), 7.0, dtype=torch.float64, device="cuda:0")        # one spare case-block behind the last case
        whip.fit_many_device(dim, order, _t(b["xk"]), _t(b["fk"]), _t(nk), _t(b["xi"]), fi, _t(b["kn"]), _t(b["wm"]), sens=guard[:n])
        torch.cuda.synchronize()
        assert bool((guard[n] == 7.0).all()), "sens written past the last case"
        res[name] = (fi.cpu().numpy(), guard[:n].cpu().numpy())
    assert np.array_equal(res["bad"][0], res["ok"][0])
    assert np.array_equal(res["bad"][1], res["ok"][1], equal_nan=True)


# ----------------------------------------------------------------------------------------------------------------------
# iterative refinement of the 10- / 15-unknown systems in the chunked tile kernel (csrc/fit_chunk.hip ITER; impl.pyx:986-1083)

@pytest.mark.parametrize("dim,order,Kn,kn", [(2, 4, 64, 1), (2, 4, 30, 0), (2, 4, 100, 0b101), (2, 4, 120, 1), (2, 3, 80, 0), (2, 3, 128, 0b10)])
def test_refinement_in_the_chunked_tile_kernel(wlsqm, oracle, dim, order, Kn, kn, monkeypatch):
    """Against the oracle's solve_iterative per column (noise-floor bound of tests/_parity.py) and against the lane-per-case
    kernel these shapes took before; ragged nk, knowns, a partial last tile; K = 120 is beyond the LDS cache of the tile (the sweeps
    re-stage their chunks)."""
    import torch
    import wlsqm.hip as whip
    n = 1000 + 7
    b = _batch(dim, order, Kn, n, Kn + kn)
    knv = np.full(n, kn, np.int64); knv[::9] = 0
    orders = np.full(n, order, np.int32)
    res = {}
    monkeypatch.setenv("WLSQM_HIP_STAGE_REFINE", "0")            # (round 4: these shapes take csrc/fit_stage_iter.hip first; this test keeps the chunked kernel covered)
    for tag in ("chunk", "lane"):
        if tag == "lane":
            monkeypatch.setenv("WLSQM_HIP_DISABLE_CHUNK_REFINE", "1")
        fi = _t(b["fi0"])
        it = whip.fit_many_device(dim, order, _t(b["xk"]), _t(b["fk"]), _t(b["nk"]), _t(b["xi"]), fi, _t(knv), _t(b["wm"]), iterative=True,
                                  max_iter=10, want_iterations=True)
        res[tag] = (fi.cpu().numpy(), it, whip.last_kernel())
    monkeypatch.delenv("WLSQM_HIP_DISABLE_CHUNK_REFINE")
    assert res["chunk"][2] == "chunk-refine" and res["lane"][2] == "lane", (res["chunk"][2], res["lane"][2])
    assert 1 <= res["chunk"][1] <= 10
    fo = b["fi0"].copy()
    oracle.fit_many(dim, b["xk"], b["fk"], b["nk"], b["xi"], fo, None, 0, orders, knv, b["wm"], iterative=True, max_iter=10, ntasks=8)
    truth = P.truth_fit(dim, b["xk"], b["fk"], b["nk"], b["xi"], b["fi0"], orders, knv, b["wm"])
    P.assert_parity(res["chunk"][0], fo, truth, "chunk-refine vs oracle")
    for a in range(b["no"]):                                    # knowns bit-identical
        sel = (knv >> a) & 1 == 1
        assert np.array_equal(res["chunk"][0][sel, a], b["fi0"][sel, a])
    # max_iter <= 0: the for/else of impl.pyx:1080-1081 returns 1 and the result is the unrefined fit
    fi0_d = _t(b["fi0"]); fi1_d = _t(b["fi0"])
    it0 = whip.fit_many_device(dim, order, _t(b["xk"]), _t(b["fk"]), _t(b["nk"]), _t(b["xi"]), fi0_d, _t(knv), _t(b["wm"]), iterative=True,
                               max_iter=0, want_iterations=True)
    assert it0 == 1 and whip.last_kernel() == "chunk-refine"
    whip.fit_many_device(dim, order, _t(b["xk"]), _t(b["fk"]), _t(b["nk"]), _t(b["xi"]), fi1_d, _t(knv), _t(b["wm"]))
    P.assert_parity(fi0_d.cpu().numpy(), fi1_d.cpu().numpy(), truth, "max_iter 0 vs basic fit")


def test_c3_iterative_vs_reference_golden_at_the_headline_density(wlsqm):
    """fit_2D_iterative_many_parallel of the reference on config_C3_1M (fi_iter of the fixture), through the reference's own
    signature: lands on the one-lane-per-case refinement kernel (round 4; the chunk-refine kernel before)."""
    import wlsqm.hip as whip
    c = K.config_dense("C3_1M")
    truth = P.truth_fit(2, c["xk"], c["fk"], c["nk_a"], c["xi"], c["fi0"], c["order_a"], c["knowns_a"], c["wm_a"])
    fi = c["fi0"].copy()
    it = wlsqm.fit_2D_iterative_many_parallel(xk=c["xk"], fk=c["fk"], nk=c["nk_a"], xi=c["xi"], fi=fi, sens=None, do_sens=0,
                                              order=c["order_a"], knowns=c["knowns_a"], weighting_method=c["wm_a"], max_iter=10)
    assert whip.last_kernel() == "stage-refine", whip.last_kernel()
    assert 1 <= it <= 10
    P.assert_parity(fi, c["g"]["fi_iter"], truth, "C3_1M iterative vs reference")


# ----------------------------------------------------------------------------------------------------------------------
# per-case polynomial orders on the device-resident API (simple.pyx:379-381 takes a per-case array)

@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("mode", ["basic", "sens", "iter"])
def test_per_case_orders_on_the_device_api(wlsqm, oracle, dim, mode):
    """The heterogeneous sweep fixture (orders 0-4 mixed, every knowns mask, ragged nk) in ONE device-resident call with an order
    TENSOR: bucketed on the device, against the reference's outputs (tests/golden/sweep_*.npz) per order with the noise-floor
    bound, the NaN / untouched pattern of sens exactly, knowns bit-identical; the same call in strict mode equals the oracle
    bit for bit."""
    import torch
    import wlsqm.hip as whip
    d = K.sweep(dim)
    n = len(d["nk"])
    do_sens, iterative = mode == "sens", mode == "iter"
    truth = P.truth_fit(dim, d["xk"], d["fk"], d["nk"], d["xi"], d["fi_in"], d["order"], d["knowns"], d["wm"])
    for strict in (False, True):
        fi = _t(d["fi_in"])
        sens = torch.full(tuple(d["sens"].shape), 777.0, dtype=torch.float64, device="cuda:0") if do_sens else None
        it = whip.fit_many_device(dim, _t(d["order"]), _t(d["xk"]), _t(d["fk"]), _t(d["nk"]), _t(d["xi"]), fi, _t(d["knowns"]), _t(d["wm"]),
                                  sens=sens, iterative=iterative, max_iter=10, want_iterations=iterative, strict=strict)
        torch.cuda.synchronize()
        fi = fi.cpu().numpy()
        ref = d["fi_iter"] if iterative else d["fi"]
        for o in range(5):
            s = d["order"] == o
            no = K.NDOF[dim][o]
            P.assert_parity(fi[s, :no], ref[s, :no], truth[s, :no], "order tensor, dim %d order %d %s strict=%s" % (dim, o, mode, strict))
            assert np.array_equal(fi[s, no:], d["fi_in"][s, no:]), "columns beyond no must stay untouched"
        for j in range(n):
            for a in range(K.NDOF[dim][int(d["order"][j])]):
                if (int(d["knowns"][j]) >> a) & 1:
                    assert fi[j, a] == d["fi_in"][j, a]
        if do_sens:
            sn = sens.cpu().numpy()
            assert np.array_equal(np.isnan(sn), np.isnan(d["sens"]))
            assert np.array_equal(sn == 777.0, d["sens"] == 777.0)
        if iterative:
            assert 1 <= it <= 10
        if strict:
            fo = d["fi_in"].copy()
            so = np.full(d["sens"].shape, 777.0) if do_sens else None
            ito = oracle.fit_many(dim, d["xk"], d["fk"], d["nk"], d["xi"], fo, so, do_sens, d["order"], d["knowns"], d["wm"],
                                  iterative=iterative, max_iter=10)
            assert np.array_equal(fi, fo), "strict + order tensor must equal the oracle bit for bit"
            if do_sens:
                assert np.array_equal(sn, so, equal_nan=True)
            if iterative:
                assert it == ito


def test_order_tensor_call_captures_into_a_hip_graph(wlsqm):
    """No host synchronisation in the order-tensor call: it can be captured, and the replay follows the buffers' current contents."""
    import torch
    import wlsqm.hip as whip
    d = K.sweep(2)
    args = [_t(d[k]) for k in ("xk", "fk", "nk", "xi")]
    fi = _t(d["fi_in"]); order = _t(d["order"]); kn = _t(d["knowns"]); wm = _t(d["wm"])
    run = lambda: whip.fit_many_device(2, order, args[0], args[1], args[2], args[3], fi, kn, wm)
    run(); torch.cuda.synchronize()
    eager = fi.clone()
    fi.copy_(_t(d["fi_in"]))
    g = torch.cuda.CUDAGraph()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            run()
    torch.cuda.synchronize()
    assert torch.equal(fi, _t(d["fi_in"])), "nothing may run during capture"
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(fi, eager)
    # other data, same graph
    args[1].mul_(2.0); fi.copy_(_t(d["fi_in"]))
    g.replay(); torch.cuda.synchronize()
    replayed = fi.clone()
    fi.copy_(_t(d["fi_in"])); run(); torch.cuda.synchronize()
    assert torch.equal(replayed, fi)


# ----------------------------------------------------------------------------------------------------------------------
# the reference's own harness shape on the index-based path (examples/wlsqm_example.py:103-133: ball query, max_nk = 100, order 4)

def test_example_harness_index_based_and_strict(wlsqm, oracle):
    """testmany2d through the device-resident INDEX-BASED entry point (hoods into the point table, ragged nk, padding = -1 never
    dereferenced), with the neighbourhoods of the golden and with those of the GPU radius search (same sets), against the
    reference's captured fi; the same call in strict mode equals the oracle bit for bit."""
    import torch
    import synth
    import wlsqm.hip as whip
    g = K.golden("testmany2d.npz")
    N = int(g["N"]); r = float(g["r"])
    S = synth.halton(N, 2); F = synth.field(S)
    hoods, nk = g["hoods"], g["nk"]
    hp = np.where(hoods >= 0, hoods, 0)
    xk = S[hp]; fk = F[hp]
    o = np.full(N, 4, np.int32); kn = np.full(N, wlsqm.b2_F, np.int64); w = np.full(N, wlsqm.WEIGHT_CENTER, np.int32)
    fi0 = np.zeros((N, 15)); fi0[:, 0] = F
    truth = P.truth_fit(2, xk, fk, nk, S, fi0, o, kn, w)
    S_d, F_d = _t(S), _t(F)
    fi = _t(fi0)
    whip.fit_cloud_device(2, 4, S_d, F_d, _t(hoods), fi, _t(nk), _t(kn), _t(w))
    torch.cuda.synchronize()
    assert whip.last_kernel() != "lane", whip.last_kernel()
    P.assert_parity(fi.cpu().numpy(), g["fi"], truth, "testmany2d index-based")
    # the GPU radius search finds the same neighbourhoods (as sets: the reference's ball query returns them unsorted)
    h_gpu, nk_gpu = whip.ball(S_d, r, 100)
    torch.cuda.synchronize()
    h_gpu, nk_gpu = h_gpu.cpu().numpy(), nk_gpu.cpu().numpy()
    assert np.array_equal(nk_gpu, nk)
    for j in range(0, N, 37):
        assert set(h_gpu[j, : nk[j]].tolist()) == set(hoods[j, : nk[j]].tolist()), j
    # strict mode on the index-based path: the oracle's bits (the oracle gets the gathered dense rows of the SAME neighbour order)
    fi_s = _t(fi0)
    whip.fit_cloud_device(2, 4, S_d, F_d, _t(hoods), fi_s, _t(nk), _t(kn), _t(w), strict=True)
    torch.cuda.synchronize()
    assert whip.last_kernel() in ("strict", "strict-rows", "strict-lane")
    fo = fi0.copy()
    oracle.fit_many(2, xk, fk, nk, S, fo, None, 0, o, kn, w, ntasks=8)
    assert np.array_equal(fi_s.cpu().numpy(), fo)


@pytest.mark.gpu
@pytest.mark.parametrize("Kn", [26, 40, 64])
@pytest.mark.parametrize("pad", [-1, "npoints"])
def test_index_based_2d_order4_runs_the_one_kernel_ring(wlsqm, oracle, Kn, pad, monkeypatch):
    """index-based 2D order 4 (the reference's harness layout, examples/wlsqm_example.py:103-133) in ONE kernel:
    the LDS ring filled by per-lane DMA gathers from the point table.  Ragged neighbourhoods with scipy-style padding, point_index,
    mixed knowns masks, a tail tile: same bits as the dense ring kernel on the gathered rows (it IS the same arithmetic once the
    ring is filled), parity with the oracle, and the padding never dereferenced."""
    import torch
    import synth
    import wlsqm.hip as whip
    monkeypatch.setenv("WLSQM_HIP_STAGE_GATHER", "0")        # (round 4: index-based input takes the gathering staged kernel by default)
    rng = np.random.default_rng(Kn)
    npts, n = 6000, 4091                                      # 255 full tiles + a tail tile of 11 cases
    S = synth.halton(npts, 2); F = synth.field(S)
    pidx = rng.permutation(npts)[:n].astype(np.int32)
    hoods = synth.knn(S, Kn, query=pidx).astype(np.int32)
    nk = rng.integers(17, Kn + 1, n).astype(np.int32); nk[::3] = Kn
    hp = hoods.copy(); hp[np.arange(Kn)[None, :] >= nk[:, None]] = npts if pad == "npoints" else pad
    kn = rng.choice(np.array([0, wlsqm.b2_F, wlsqm.b2_F | wlsqm.b2_X2, 1 << 14], np.int64), n)
    w = rng.choice(np.array([wlsqm.WEIGHT_UNIFORM, wlsqm.WEIGHT_CENTER], np.int32), n)
    fi0 = rng.uniform(-1, 1, (n, 15)); fi0[:, 0] = F[pidx]
    fi = _t(fi0)
    whip.fit_cloud_device(2, 4, _t(S), _t(F), _t(hp), fi, _t(nk), _t(kn), _t(w), point_index=_t(pidx))
    torch.cuda.synchronize()
    assert whip.last_kernel() == "tile-solve-gather", whip.last_kernel()
    got = fi.cpu().numpy()
    hc = np.where(np.arange(Kn)[None, :] < nk[:, None], hoods, 0).astype(np.int64)
    xk, fk, xi = S[hc], F[hc], S[pidx]
    fd = _t(fi0)
    os.environ["WLSQM_HIP_STAGE"] = "0"                      # the dense RING (round 4: dense input takes the staged kernel by default)
    try:
        whip.fit_many_device(2, 4, _t(xk), _t(fk), _t(nk), _t(xi), fd, _t(kn), _t(w))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("WLSQM_HIP_STAGE", None)
    assert whip.last_kernel() == "tile-solve", whip.last_kernel()
    assert np.array_equal(got.view(np.int64), fd.cpu().numpy().view(np.int64)), "gathered ring != dense ring on the same rows"
    o = np.full(n, 4, np.int32)
    ref = fi0.copy()
    oracle.fit_many(2, xk, fk, nk, xi, ref, None, 0, o, kn, w, ntasks=8)
    truth = P.truth_fit(2, xk, fk, nk, xi, fi0, o, kn, w)
    P.assert_parity(got, ref, truth, "index-based ring K = %d" % Kn)

