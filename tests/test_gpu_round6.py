"""GPU tests added in round 6: ragged neighbourhoods (the wave-uniform chunk count of the staged kernels, the host path's neighbour-count
order), the accurate mode's one-launch form is covered by test_gpu_accurate.py."""
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


def _ragged(dim, order, Kn, n, seed, runs):
    """Rows of Kn slots; the neighbour count of a case is drawn per RUN of 64 consecutive cases (`runs`: per-run (lo, hi) ranges cycled over the
    batch), so that some waves hold short cases only, some full ones only and some a mixture; padding slots are NaN (never read:
    simple.pyx:147)."""
    rng = np.random.default_rng(seed)
    no = K.NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim))
    xk = xi[:, None, :] + 0.05 * rng.uniform(-1, 1, (n, Kn, dim))
    nk = np.empty(n, np.int32)
    for g0 in range(0, n, 64):
        lo, hi = runs[(g0 // 64) % len(runs)]
        nk[g0:g0 + 64] = rng.integers(lo, hi + 1, min(64, n - g0))
    # neighbours sorted by distance among the valid ones (a k-nearest-neighbour search's rows) for every second run, unsorted for the others
    for j in range(n):
        if (j // 64) % 2 == 0:
            d2 = ((xk[j, :nk[j]] - xi[j]) ** 2).sum(-1)
            xk[j, :nk[j]] = xk[j, :nk[j]][np.argsort(d2)]
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
    pad = np.arange(Kn)[None, :] >= nk[:, None]
    xk[pad] = np.nan; fk[pad] = np.nan
    fi0 = np.zeros((n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
    return dict(xi=xi, xk=xk, fk=fk, nk=nk, fi0=fi0, no=no)


@pytest.mark.parametrize("dim,order,Kn,kn", [(2, 4, 100, 1), (2, 4, 64, 0), (2, 2, 32, 0), (3, 2, 40, 0), (2, 3, 48, 1), (3, 3, 64, 0), (3, 4, 96, 0)])
def test_waves_stage_the_chunks_their_own_cases_need(wlsqm, oracle, dim, order, Kn, kn):
    """a wave of the staged kernels moves ceil(max over ITS cases of nk / 8) chunks, not the row's.  Batches whose 64-case
    groups are all short / all full / mixed / one neighbour more than a chunk boundary, sorted and unsorted rows, padding NaN: the usual bound
    against the CPU port, bit-identical to the same cases in a batch of full-length groups' company (a case's bits do not depend on what its
    wave stages), and the index-based form agrees."""
    import torch
    import wlsqm.hip as whip
    no = K.NDOF[dim][order]
    lo = no + 4
    runs = [(lo, min(Kn, lo + 6)), (Kn, Kn), (lo, Kn), (min(Kn, 8 * ((lo + 7) // 8) + 1),) * 2, (Kn // 2, Kn // 2 + 3)]
    n = 64 * 23 + 17
    b = _ragged(dim, order, Kn, n, 100 * dim + order, runs)
    kna = np.full(n, kn, np.int64); wm = np.full(n, wlsqm.WEIGHT_CENTER, np.int32); oa = np.full(n, order, np.int32)
    ref = b["fi0"].copy()
    xk0 = np.nan_to_num(b["xk"]); fk0 = np.nan_to_num(b["fk"])           # (the CPU port reads only k < nk as well; NaN-free copies for the extended-precision solve)
    oracle.fit_many(dim, xk0, fk0, b["nk"], b["xi"], ref, None, 0, oa, kna, wm, ntasks=8)
    truth = P.truth_fit(dim, xk0, fk0, b["nk"], b["xi"], b["fi0"], oa, kna, wm)
    outs = {}
    for hint in ("ragged", None, "full"):                              # wlsqm_hip_set_row_hint: the caller's word, the kernels finding out, the default
        fi = _t(b["fi0"])
        with whip.row_hint(hint):
            whip.fit_many_device(dim, order, _t(b["xk"]), _t(b["fk"]), _t(b["nk"]), _t(b["xi"]), fi, _t(kna), _t(wm))
            torch.cuda.synchronize()
            kern = whip.last_kernel()
        outs[hint] = fi.cpu().numpy()
        if (dim, order) != (3, 4):
            assert kern == ("stage-ragged" if hint == "ragged" else "stage"), (hint, kern)
    got = outs["ragged"]
    assert not np.isnan(got).any(), "a padding slot was read"
    for hint in (None, "full"):
        assert np.array_equal(outs[hint].view(np.int64), got.view(np.int64)), "the hint changes the bits (%s)" % hint
    P.assert_parity(got, ref, truth, "ragged waves, %dD order %d" % (dim, order))
    # the same cases in another company: every second group replaced by full-length cases
    sel = np.nonzero((np.arange(n) // 64) % 2 == 0)[0]
    b2 = _ragged(dim, order, Kn, n, 7, [(Kn, Kn)])
    for k in ("xi", "xk", "fk", "nk", "fi0"):
        b2[k][sel] = b[k][sel]
    fi2 = _t(b2["fi0"])
    with whip.row_hint(None):
        whip.fit_many_device(dim, order, _t(b2["xk"]), _t(b2["fk"]), _t(b2["nk"]), _t(b2["xi"]), fi2, _t(kna), _t(wm))
        torch.cuda.synchronize()
    assert np.array_equal(fi2.cpu().numpy()[sel].view(np.int64), got[sel].view(np.int64)), "a case's bits depend on its batch-mates"


def test_host_path_packs_ragged_batches_in_neighbour_count_order(wlsqm, oracle, monkeypatch):
    """The host entry points stage a ragged uniform-order batch in neighbour-count order (csrc/api.hip) and scatter the results back: the
    same bits as with WLSQM_HIP_HOST_NK_ORDER=0, fi and sensitivities, user rows with strides, known DOFs untouched, fk aliasing fi[:, 0]."""
    rng = np.random.default_rng(3)
    n, Kn, dim, order = 3000, 60, 2, 4
    no = K.NDOF[dim][order]
    b = _ragged(dim, order, Kn, n, 11, [(no + 3, Kn)])
    b["nk"] = rng.integers(no + 12, Kn + 1, n).astype(np.int32)         # every case its own count (well away from a determined fit: its noise is not the subject)
    b["xk"] = np.nan_to_num(b["xk"]); b["fk"] = np.nan_to_num(b["fk"])
    kna = rng.choice(np.array([0, 1, 1, 5], np.int64), n); wm = rng.choice(np.array([1, 2], np.int32), n); oa = np.full(n, order, np.int32)
    wide = np.full((n, no + 3), 777.0); wide[:, :no] = rng.uniform(-1, 1, (n, no)); wide[:, 0] = b["fi0"][:, 0]

    def run():
        fi = wide.copy()
        sens = np.full((n, Kn, no), 777.0)
        rc = wlsqm.fit_2D_many(xk=b["xk"], fk=b["fk"], nk=b["nk"], xi=b["xi"], fi=fi, sens=sens, do_sens=1, order=oa, knowns=kna, weighting_method=wm)
        assert rc == 0
        return fi, sens
    fi_a, sens_a = run()
    monkeypatch.setenv("WLSQM_HIP_HOST_NK_ORDER", "0")
    fi_b, sens_b = run()
    assert np.array_equal(fi_a.view(np.int64), fi_b.view(np.int64)) and np.array_equal(sens_a, sens_b, equal_nan=True)
    assert np.array_equal(fi_a[:, no:], wide[:, no:])                    # beyond the DOFs: untouched
    known = np.array([[(int(k) >> a) & 1 for a in range(no)] for k in kna], bool)
    assert np.array_equal(fi_a[:, :no][known], wide[:, :no][known])
    ref = np.ascontiguousarray(wide[:, :no]).copy()
    oracle.fit_many(dim, b["xk"], b["fk"], b["nk"], b["xi"], ref, None, 0, oa, kna, wm, ntasks=8)
    truth = P.truth_fit(dim, b["xk"], b["fk"], b["nk"], b["xi"], np.ascontiguousarray(wide[:, :no]), oa, kna, wm)
    P.assert_parity(fi_a[:, :no], ref, truth, "host path, ragged batch in neighbour-count order")


def test_unknown_row_hint_across_streams_and_in_a_graph(wlsqm, oracle):
    """wlsqm_hip_set_row_hint(0): the plain kernel marks its short waves in per-group status bytes and the RAGGED copy runs them behind it.
    The bytes live in a buffer that persists per stream (handed out under a lock held across the two enqueues) — or in stream-ordered
    scratch inside a graph capture: calls alternating between two streams and a captured, replayed call return the bits of a plain call."""
    import torch
    import wlsqm.hip as whip
    dim, order, Kn, n = 2, 4, 64, 64 * 40 + 5
    no = K.NDOF[dim][order]
    b = _ragged(dim, order, Kn, n, 77, [(no + 6, 30), (Kn, Kn), (no + 6, Kn)])
    kna = np.ones(n, np.int64); wm = np.full(n, wlsqm.WEIGHT_CENTER, np.int32)
    args = [_t(b[k]) for k in ("xk", "fk", "nk", "xi")]
    kn_d, wm_d = _t(kna), _t(wm)
    fi = _t(b["fi0"])
    whip.fit_many_device(dim, order, *args, fi, kn_d, wm_d)             # the default hint: the plain kernel for every wave
    torch.cuda.synchronize()
    want = fi.cpu().numpy().view(np.int64)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    with whip.row_hint(None):
        for rep in range(4):
            for st in (s1, s2):
                f2 = _t(b["fi0"])
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    whip.fit_many_device(dim, order, *args, f2, kn_d, wm_d)
                outs.append(f2)
        torch.cuda.synchronize()
        for f2 in outs:
            assert np.array_equal(f2.cpu().numpy().view(np.int64), want)
        fg = _t(b["fi0"])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s1):
            whip.fit_many_device(dim, order, *args, fg, kn_d, wm_d)
        for rep in range(3):
            fg.copy_(_t(b["fi0"]))
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(fg.cpu().numpy().view(np.int64), want), "replay %d" % rep


def test_expert_solver_picks_the_form_for_its_rows_at_prepare(wlsqm):
    """ExpertSolver.prepare() has the rows in the caller's memory: it looks at 64 of them, and solve() runs the staged kernel's form for
    unsorted rows when most are out of distance order (prepare_device: the caller's word, wlsqm.hip.row_hint(sorted=...)).  The same bits
    as the one-shot call on the same cases either way."""
    import torch
    import wlsqm.hip as whip
    dim, order, Kn, n = 2, 3, 48, 64 * 30 + 9                             # (the two forms exist for the dense systems of up to 10 unknowns)
    no = K.NDOF[dim][order]
    rng = np.random.default_rng(5)
    xi = rng.uniform(0, 1, (n, dim))
    off = 0.05 * rng.uniform(-1, 1, (n, Kn, dim))
    srt = np.take_along_axis(off, np.argsort((off ** 2).sum(-1), axis=1)[..., None], axis=1)
    nk = np.full(n, Kn, np.int32); kna = np.ones(n, np.int64); wm = np.full(n, wlsqm.WEIGHT_CENTER, np.int32); oa = np.full(n, order, np.int32)
    for name, o, kern in (("shuffled", off, "stage-own"), ("sorted", srt, "stage")):
        xk = xi[:, None, :] + o
        fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
        fi0 = np.zeros((n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, 1])
        fi_d = _t(fi0)
        with whip.row_hint(sorted=(name == "sorted")):
            whip.fit_many_device(dim, order, _t(xk), _t(fk), _t(nk), _t(xi), fi_d, _t(kna), _t(wm))
            torch.cuda.synchronize()
        want = fi_d.cpu().numpy().view(np.int64)
        s = wlsqm.ExpertSolver(dimension=dim, nk=nk, order=oa, knowns=kna, weighting_method=wm, algorithm=wlsqm.ALGO_BASIC, do_sens=False, ntasks=1)
        s.prepare(xi=xi, xk=xk)
        fi = fi0.copy()
        s.solve(fk=fk, fi=fi)
        assert whip.last_kernel() == kern, (name, whip.last_kernel())
        assert np.array_equal(fi.view(np.int64), want), name
        s2 = wlsqm.ExpertSolver(dimension=dim, nk=nk, order=oa, knowns=kna, weighting_method=wm, algorithm=wlsqm.ALGO_BASIC, do_sens=False, ntasks=1)
        with whip.row_hint(sorted=(name == "sorted")):
            s2.prepare_device(_t(xi), _t(xk))
        fi2 = _t(fi0)
        s2.solve_device(_t(fk), fi2)
        torch.cuda.synchronize()
        assert whip.last_kernel() == kern, (name, whip.last_kernel())
        assert np.array_equal(fi2.cpu().numpy().view(np.int64), want), name
