"""Pins the CPU oracle (oracle/wlsqm_oracle.c) against golden vectors captured from the real
reference (tests/golden/make_golden.py).  CPU only.  If these pass, oracle == reference up to the
LAPACK rounding order, and the GPU parity tests may use the oracle as the checker at any size."""
import numpy as np
import pytest

import _cases as K
import _parity as P
from oracle import oracle

EPS = np.finfo(np.float64).eps


def test_number_of_dofs_table():
    # reference tests/test_package.py:47-53
    for dim in (1, 2, 3):
        assert [oracle.number_of_dofs(dim, k) for k in range(5)] == K.NDOF[dim]
    assert oracle.number_of_dofs(4, 2) == -1 and oracle.number_of_dofs(2, 5) == -2   # infra.pyx:68-73


def test_remap_bit_exact():
    tab = K.golden("remap.npz")["table"]
    for row in tab:
        n, mask, nr = int(row[0]), int(row[1]), int(row[2])
        k, o2r, r2o = oracle.remap(n, mask)
        assert k == nr
        assert np.array_equal(o2r, row[3:3 + n]) and np.array_equal(r2o, row[38:38 + n])


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_sweep_intermediates_and_fi(dim):
    d = K.sweep(dim)
    fi = d["fi_in"].copy()
    rc, cap = oracle.fit_many(dim, d["xk"], d["fk"], d["nk"], d["xi"], fi, None, 0, d["order"], d["knowns"], d["wm"],
                              debug_capture=True)
    assert rc == 0
    # index work: bit-exact
    for k in ("o2r", "r2o", "ipiv"):
        assert np.array_equal(cap[k], d[k]), k
    # arithmetic before the LAPACK call: bit-exact with the reference build
    for k in ("c", "w", "A", "row_scale", "col_scale"):
        assert np.array_equal(cap[k], d[k]), k
    # (fi itself is judged in test_sweep_noise_floor_parity: LU/solution differ by LAPACK's rounding order)
    n = len(d["nk"])
    # knowns untouched (bit-identical), columns beyond `no` untouched
    for j in range(n):
        no = K.NDOF[dim][int(d["order"][j])]
        for a in range(no):
            if (int(d["knowns"][j]) >> a) & 1:
                assert fi[j, a] == d["fi_in"][j, a]
        assert np.array_equal(fi[j, no:], d["fi_in"][j, no:])


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_sweep_noise_floor_parity(dim):
    """Same sweep, judged with the extended-precision noise floor, grouped by order."""
    d = K.sweep(dim)
    fi = d["fi_in"].copy()
    oracle.fit_many(dim, d["xk"], d["fk"], d["nk"], d["xi"], fi, None, 0, d["order"], d["knowns"], d["wm"])
    truth = P.truth_fit(dim, d["xk"], d["fk"], d["nk"], d["xi"], d["fi_in"], d["order"], d["knowns"], d["wm"])
    for o in range(5):
        sel = d["order"] == o
        no = K.NDOF[dim][o]
        P.assert_parity(fi[sel, :no], d["fi"][sel, :no], truth[sel, :no], "sweep dim %d order %d" % (dim, o))


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_sweep_sens(dim):
    d = K.sweep(dim)
    fi = d["fi_in"].copy()
    sens = np.full(d["sens"].shape, 777.0)
    oracle.fit_many(dim, d["xk"], d["fk"], d["nk"], d["xi"], fi, sens, 1, d["order"], d["knowns"], d["wm"])
    assert np.array_equal(np.isnan(sens), np.isnan(d["sens"]))
    assert np.array_equal(sens == 777.0, d["sens"] == 777.0)          # never-written entries stay untouched
    for j in range(len(d["nk"])):
        no = K.NDOF[dim][int(d["order"][j])]
        kappa = K.scaled_cond(d, j, no, d["knowns"][j])
        a, b = sens[j], d["sens"][j]
        m = ~np.isnan(b) & (b != 777.0)
        if m.any():
            assert np.abs(a[m] - b[m]).max() <= (1e-10 + 1e3 * kappa * EPS) * np.abs(b[m]).max(), (j, kappa)


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_sweep_iterative(dim):
    d = K.sweep(dim)
    fi = d["fi_in"].copy()
    it = oracle.fit_many(dim, d["xk"], d["fk"], d["nk"], d["xi"], fi, None, 0, d["order"], d["knowns"], d["wm"],
                         iterative=True, max_iter=10)
    assert it == int(d["iters"])
    truth = P.truth_fit(dim, d["xk"], d["fk"], d["nk"], d["xi"], d["fi_in"], d["order"], d["knowns"], d["wm"])
    for o in range(5):
        sel = d["order"] == o
        no = K.NDOF[dim][o]
        P.assert_parity(fi[sel, :no], d["fi_iter"][sel, :no], truth[sel, :no], "iter sweep dim %d order %d" % (dim, o))


@pytest.mark.parametrize("name", K.CONFIGS)
def test_config_parity(name):
    c = K.config(name)
    g = c["g"]
    truth = P.truth_fit(c["dim"], c["xk"], c["fk"], c["nk_a"], c["xi"], c["fi0"], c["order_a"], c["knowns_a"], c["wm_a"])
    fi = c["fi0"].copy()
    oracle.fit_many(c["dim"], c["xk"], c["fk"], c["nk_a"], c["xi"], fi, None, 0, c["order_a"], c["knowns_a"], c["wm_a"],
                    ntasks=4)
    P.assert_parity(fi, g["fi"], truth, name)
    fi2 = c["fi0"].copy()
    it = oracle.fit_many(c["dim"], c["xk"], c["fk"], c["nk_a"], c["xi"], fi2, None, 0, c["order_a"], c["knowns_a"],
                         c["wm_a"], iterative=True, max_iter=10, ntasks=4)
    assert it == int(g["iters"])
    P.assert_parity(fi2, g["fi_iter"], truth, name + " iterative")
    ns = g["sens"].shape[0]
    fi3 = c["fi0"][:ns].copy()
    sens = np.zeros(g["sens"].shape)
    oracle.fit_many(c["dim"], c["xk"][:ns], c["fk"][:ns], c["nk_a"][:ns], c["xi"][:ns], fi3, sens, 1,
                    c["order_a"][:ns], c["knowns_a"][:ns], c["wm_a"][:ns])
    m = ~np.isnan(g["sens"])
    assert np.array_equal(np.isnan(sens), ~m)
    assert np.abs(sens[m] - g["sens"][m]).max() <= 1e-10 * np.abs(g["sens"][m]).max()


def test_edge_iter_quirk_and_stencil():
    e = K.golden("edge.npz")
    xk, fk = e["iter_xk"], e["iter_fk"]
    for mi in (0, 1, 2, 10):
        fi = np.zeros((1, 6))
        it = oracle.fit_many(2, xk[None], fk[None], np.array([20], np.int32), np.zeros((1, 2)), fi, None, 0,
                             np.array([2], np.int32), np.array([0], np.int64), np.array([2], np.int32),
                             iterative=True, max_iter=mi)
        ref_it = int(e["iter_mi%d_it" % mi])
        if mi <= 1:
            assert it == ref_it == 1                            # for/else quirk (impl.pyx:1080-1081): max_iter=0 returns 1
        else:
            # the stop test is exact fp equality of two residual norms (impl.pyx:1057): the count depends on the
            # last bits of LAPACK's rounding order, so only its range is implementation-independent
            assert 1 <= it <= mi and 1 <= ref_it <= mi
        assert np.allclose(fi[0], e["iter_mi%d_fi" % mi], rtol=1e-12, atol=1e-13)
    fi = np.zeros((1, 6))
    oracle.fit_many(2, e["stencil_xk"][None], e["stencil_fk"][None], np.array([5], np.int32), np.zeros((1, 2)), fi,
                    None, 0, np.array([2], np.int32), np.array([1 << 4], np.int64), np.array([1], np.int32))
    assert fi[0, 4] == 0.0                                      # known b2_XY stays bit-identical (test_stencil.py:145)
    assert np.allclose(fi[0], e["stencil_fi"], rtol=1e-11, atol=1e-12)


def test_edge_strided_views():
    e = K.golden("edge.npz")
    n, nk = 12, 14
    xkv = e["strided_big_xk"][::2, ::2, :]; fkv = e["strided_big_fk"][::2, ::2]
    big_fi = np.zeros((2 * n, 8)); fiv = big_fi[::2, :6]
    o = np.full(2 * n, 2, np.int32)[::2]; kn = np.zeros(2 * n, np.int64)[::2]
    w = np.full(2 * n, 2, np.int32)[::2]; nka = np.full(2 * n, nk, np.int32)[::2]
    oracle.fit_many(2, xkv, fkv, nka, e["strided_xi"], fiv, None, 0, o, kn, w)
    assert np.allclose(big_fi, e["strided_big_fi_after"], rtol=1e-11, atol=1e-13)
    assert np.all(big_fi[1::2] == 0) and np.all(big_fi[:, 6:] == 0)   # nothing outside the view is touched


def test_example_harness_ragged_radius_neighbourhoods():
    """examples/wlsqm_example.py testmany2d pattern (ragged nk from a radius query, order 4, F known)."""
    import synth
    g = K.golden("testmany2d.npz")
    N = int(g["N"])
    S = synth.halton(N, 2); F = synth.field(S)
    hoods, nk = g["hoods"], g["nk"]
    hp = np.where(hoods >= 0, hoods, 0)
    xk = S[hp]; fk = F[hp]
    xk[hoods < 0] = np.nan; fk[hoods < 0] = np.nan
    o = np.full(N, 4, np.int32); kn = np.ones(N, np.int64); w = np.full(N, 2, np.int32)
    fi0 = np.zeros((N, 15)); fi0[:, 0] = F
    fi = fi0.copy()
    oracle.fit_many(2, xk, fk, nk, S, fi, None, 0, o, kn, w, ntasks=4)
    truth = P.truth_fit(2, xk, fk, nk, S, fi0, o, kn, w)
    P.assert_parity(fi, g["fi"], truth, "testmany2d")


@pytest.mark.parametrize("name", K.DENSE)
def test_dense_density_oracle_vs_reference(name):
    """The oracle against the reference at the density the metric is quoted on (1M / 16M-point clouds, every 977th case):
    this is the floor two correct fp64 implementations of the SAME algorithm differ by (only LAPACK's summation order
    changes) — the number the GPU tests and bench.py print beside gpu-vs-reference."""
    c = K.config_dense(name)
    fi = c["fi0"].copy()
    oracle.fit_many(c["dim"], c["xk"], c["fk"], c["nk_a"], c["xi"], fi, None, 0, c["order_a"], c["knowns_a"], c["wm_a"])
    truth = P.truth_fit(c["dim"], c["xk"], c["fk"], c["nk_a"], c["xi"], c["fi0"], c["order_a"], c["knowns_a"], c["wm_a"])
    kn = int(c["knowns_a"][0])
    known_cols = [a for a in range(c["no"]) if (kn >> a) & 1]
    for a in known_cols:
        assert np.array_equal(fi[:, a], c["fi0"][:, a])
    acc = P.accounting(fi, c["g"]["fi"], truth=truth, conds=c["conds"], known_cols=known_cols)
    print("%s oracle-vs-reference: E_max %.2e, N_max %.2e, strict columns %d/%d" %
          (name, acc["E_max"], acc["N_max"], acc["strict_1e-10_columns"], acc["columns"]))
    assert acc["within_1e-10_plus_8N"], acc
    assert not acc["resolved_columns_missing_strict"], acc


# ---- the CPU statement of the ACCURATE numerics mode (oracle/variants.c with V_SYM), round 4 ----

@pytest.mark.parametrize("name", ["C2_1M", "C5_1M", "C5_16M"])
def test_accurate_mode_statement_against_the_reference_goldens(name):
    """variants.c without a switch IS the oracle (bit for bit); with V_SYM — the normal matrix assembled from its upper triangle and
    mirrored, everything else the reference's arithmetic: the accurate mode of the HIP library, which the GPU tests hold to this
    routine bit for bit — it stays within 1e-10 of the REFERENCE's own output on every column with a factor of two to spare, at
    the density the metric is quoted on."""
    import _cases as K
    import _parity as P
    from oracle import oracle as O
    c = K.config_dense(name)
    args = (c["dim"], c["order"], np.ascontiguousarray(c["xk"]), np.ascontiguousarray(c["fk"]), c["nk_a"], np.ascontiguousarray(c["xi"]))
    fi_o = c["fi0"].copy()
    O.fit_many(c["dim"], c["xk"], c["fk"], c["nk_a"], c["xi"], fi_o, None, 0, c["order_a"], c["knowns_a"], c["wm_a"], ntasks=8)
    fi_0 = np.ascontiguousarray(c["fi0"].copy())
    O.variant_fit_many(*args, fi_0, c["knowns_a"], c["wm_a"], flags=0)
    assert np.array_equal(fi_0, fi_o), "variants.c without a switch must be the oracle"
    fi_s = np.ascontiguousarray(c["fi0"].copy())
    O.variant_fit_many(*args, fi_s, c["knowns_a"], c["wm_a"], flags=O.V_SYM)
    E = P.column_metric(fi_s, c["g"]["fi"])
    assert np.all(E <= 0.5e-10), E


@pytest.mark.parametrize("dim,order,Kn", [(2, 2, 32), (3, 2, 40), (2, 3, 36), (2, 1, 12)])
def test_accurate_mode_statement_with_known_dofs(dim, order, Kn):
    """Round 5: the accurate kernels take cases WITH known DOFs (any mask inside the polynomial's DOFs; knowns eliminated term by term as
    impl.pyx:792-823).  variants.c without a switch still equals the oracle bit for bit on such batches (ragged nk, both weightings);
    with V_SYM it stays within 0.5e-10 of the oracle on every unknown column, and never writes a known DOF."""
    import _cases as K
    import _parity as P
    from oracle import oracle as O
    rng = np.random.default_rng(100 * dim + order)
    n, no = 600, K.NDOF[dim][order]
    xi = rng.uniform(0, 1, (n, dim))
    xk = xi[:, None, :] + 0.05 * rng.uniform(-1, 1, (n, Kn, dim))
    fk = np.sin(3 * xk[..., 0]) * np.cos(2 * xk[..., -1])
    nk = rng.integers(no + 3, Kn + 1, n).astype(np.int32)
    kn = rng.choice(np.array([1, 1, 2, 5, (1 << (no - 1)) | 2], np.int64), n)
    wm = rng.choice(np.array([1, 2], np.int32), n)
    fi0 = rng.uniform(-1, 1, (n, no)); fi0[:, 0] = np.sin(3 * xi[:, 0]) * np.cos(2 * xi[:, -1])
    ora = fi0.copy()
    O.fit_many(dim, xk, fk, nk, xi, ora, None, 0, np.full(n, order, np.int32), kn, wm, ntasks=8)
    v0 = fi0.copy(); O.variant_fit_many(dim, order, xk, fk, nk, xi, v0, kn, wm, flags=0)
    assert np.array_equal(v0.view(np.int64), ora.view(np.int64)), "variants.c without a switch must be the oracle"
    vs = fi0.copy(); O.variant_fit_many(dim, order, xk, fk, nk, xi, vs, kn, wm, flags=O.V_SYM)
    known = ((kn[:, None] >> np.arange(no)[None, :]) & 1).astype(bool)
    assert np.array_equal(vs[known].view(np.int64), fi0[known].view(np.int64)), "a known DOF was written"
    for m in range(no):
        sel = ~known[:, m]
        if sel.sum() > 0:
            e = np.abs(vs[sel, m] - ora[sel, m]).max() / np.abs(ora[sel, m]).max()
            assert e <= 0.5e-10, (m, e)


def test_oracle_on_the_prepare_once_time_levels_of_configs3():
    """tests/golden/config_C4_1M.npz: the reference's ExpertSolver prepared once and solved for four time levels (BASELINE
    configs[3]'s pattern, expert.pyx:309-655).  The oracle, one fit per level, agrees to LAPACK's rounding (column metric 1e-10)."""
    import _cases as K
    import _parity as P
    from oracle import oracle as O
    c = K.config_c4()
    for t in range(c["nlevels"]):
        fi = c["fi0"][t].copy()
        O.fit_many(2, c["xk"], c["fk"][t], c["nk_a"], c["xi"], fi, None, 0, c["order_a"], c["knowns_a"], c["wm_a"], ntasks=8)
        E = P.column_metric(fi, c["fi_ref"][t])
        assert np.all(E <= 1e-10), (t, E)
