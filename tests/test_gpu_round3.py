"""GPU tests added in round 3 (beside test_gpu_strict.py and test_gpu_rccl.py): the repack / gather scratch in bounded slices
(ADVICE r2), the logical neighbour count behind a repack, per-case order tensors on the device-resident API."""
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
