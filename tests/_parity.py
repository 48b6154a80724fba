"""Parity helpers shared by the CPU and GPU tests.

`truth_fit` is an independent extended-precision (x87 long double, eps = 5.4e-20) statement of
the WLSQM fit in numpy: it gives the noise floor of ANY fp64 implementation on a given input, so a
parity tolerance can be tied to what double precision can resolve at all.

Parity metric: per DOF column m over the batch,
    E_m = max_j |fi[j,m] - ref[j,m]| / max_j |ref[j,m]|.
The north-star tolerance is E_m <= 1e-10.  Where the fit itself is ill-conditioned in fp64 (tiny
neighbourhoods, high derivatives: the REFERENCE's own error against the extended-precision
solution, N_m, exceeds 1e-10) no two fp64 implementations can agree to 1e-10 unless they replay
each other's roundoff, so the asserted bound is  E_m <= 1e-10 + NOISE_MULT * N_m  and the test
additionally requires the candidate to be no further from the truth than the reference is (same
multiplier).  Known DOFs must be bit-identical to the input.
"""
import numpy as np

TOL = 1e-10
NOISE_MULT = 8.0

_P2 = [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2), (3, 0), (2, 1), (1, 2), (0, 3),
       (4, 0), (3, 1), (2, 2), (1, 3), (0, 4)]
_P3 = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1),
       (2, 0, 0), (1, 1, 0), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 1),
       (3, 0, 0), (2, 1, 0), (1, 2, 0), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3), (1, 0, 2), (2, 0, 1), (1, 1, 1),
       (4, 0, 0), (3, 1, 0), (2, 2, 0), (1, 3, 0), (0, 4, 0), (0, 3, 1), (0, 2, 2), (0, 1, 3), (0, 0, 4),
       (1, 0, 3), (2, 0, 2), (3, 0, 1), (2, 1, 1), (1, 2, 1), (1, 1, 2)]
_FACT = [1.0, 1.0, 2.0, 6.0, 24.0]
_NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}


def exponents(dim, order):
    no = _NDOF[dim][order]
    if dim == 1:
        return [(a,) for a in range(no)]
    return (_P2 if dim == 2 else _P3)[:no]


def truth_fit(dim, xk, fk, nk, xi, fi_in, order, knowns, wm):
    """Extended-precision WLSQM fit of a batch (same argument meaning as fit_*D_many).  1D: xk (n,K),
    xi (n,).  Returns fi (n, max_no) float64 (knowns copied from fi_in).  Independent of oracle/."""
    LD = np.longdouble
    n = len(nk)
    out = np.array(fi_in, dtype=np.float64, copy=True)
    for j in range(n):
        o, nkj, kn = int(order[j]), int(nk[j]), int(knowns[j])
        ex = exponents(dim, o)
        no = len(ex)
        unknown = [a for a in range(no) if not (kn >> a) & 1]
        extra = bin(kn >> no).count("1")                      # infra.pyx:119-121 quirk: stray high bits drop unknowns
        if extra:
            unknown = unknown[:max(len(unknown) - extra, 0)]
        if not unknown:
            continue
        if dim == 1:
            d = (np.asarray(xk[j, :nkj], LD) - LD(xi[j]))[:, None]
        else:
            d = np.asarray(xk[j, :nkj, :dim], LD) - np.asarray(xi[j, :dim], LD)[None, :]
        Cm = np.ones((nkj, no), LD)
        for a, e in enumerate(ex):
            for m, p in enumerate(e):
                if p:
                    Cm[:, a] *= d[:, m] ** p / LD(_FACT[p])
        if int(wm[j]) == 1:
            w = np.ones(nkj, LD)
        else:
            d2 = (d * d).sum(axis=1)
            t = LD(1) - np.sqrt(d2 / d2.max())
            w = LD(1e-4) + (LD(1) - LD(1e-4)) * t * t
        f = np.asarray(fk[j, :nkj], LD).copy()
        dropped = [a for a in range(no) if not (kn >> a) & 1 and a not in unknown]
        for a in range(no):
            if (kn >> a) & 1:
                f -= Cm[:, a] * LD(fi_in[j, a])               # known DOFs move to the right-hand side
        Cu = Cm[:, unknown]
        # column equilibration (exact in effect; keeps the elimination well scaled), then normal equations
        s = np.sqrt((w[:, None] * Cu * Cu).sum(axis=0))
        s[s == 0] = 1
        Cs = Cu / s
        A = (Cs * w[:, None]).T @ Cs
        b = (Cs * w[:, None]).T @ f
        x = _solve_ld(A, b) / s
        out[j, unknown] = x.astype(np.float64)
        del dropped
    return out


def _solve_ld(A, b):
    """Gaussian elimination with partial pivoting in long double."""
    A = A.copy(); b = b.copy()
    n = len(b)
    for c in range(n):
        p = c + int(np.argmax(np.abs(A[c:, c])))
        if p != c:
            A[[c, p]] = A[[p, c]]; b[[c, p]] = b[[p, c]]
        piv = A[c, c]
        for r in range(c + 1, n):
            m = A[r, c] / piv
            if m != 0:
                A[r, c:] -= m * A[c, c:]
                b[r] -= m * b[c]
    x = np.zeros(n, np.longdouble)
    for r in range(n - 1, -1, -1):
        x[r] = (b[r] - A[r, r + 1:] @ x[r + 1:]) / A[r, r]
    return x


def column_metric(a, ref):
    """E_m per column; columns whose reference is identically zero compare absolutely."""
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    scale = np.nanmax(np.abs(ref), axis=0)
    scale = np.where(scale > 0, scale, 1.0)
    with np.errstate(invalid="ignore"):
        return np.nanmax(np.abs(a - ref), axis=0) / scale


def assert_parity(cand, ref, truth=None, what="", tol=TOL, noise_mult=NOISE_MULT):
    """Assert the column metric; with `truth`, allow the reference's own fp64 noise floor."""
    cand = np.asarray(cand); ref = np.asarray(ref)
    assert cand.shape == ref.shape, (cand.shape, ref.shape)
    if not np.array_equal(np.isnan(cand), np.isnan(ref)):
        where = np.argwhere(np.isnan(cand) != np.isnan(ref))[:6]
        raise AssertionError("%s: NaN pattern differs at %s: candidate %s, reference %s"
                             % (what, where.tolist(), [cand[tuple(w)] for w in where], [ref[tuple(w)] for w in where]))
    E = column_metric(cand, ref)
    bound = np.full_like(E, tol)
    if truth is not None:
        N = column_metric(ref, truth)
        bound = tol + noise_mult * N
        Ec = column_metric(cand, truth)
        assert np.all(Ec <= tol + noise_mult * N), (
            "%s: candidate further from the extended-precision solution than the reference allows: %s vs noise %s"
            % (what, Ec, N))
    assert np.all(E <= bound), "%s: column metric %s exceeds bound %s" % (what, E, bound)
    return E


COND_EDGES = (1.0, 1e1, 1e2, 1e3, 1e4, 1e5)     # six bins: [1, 10), [10, 1e2), ..., [1e5, inf)


def accounting(cand, ref, truth=None, oracle=None, conds=None, known_cols=()):
    """Strict-tolerance accounting of one batch against the REFERENCE's output (SURVEY.md section 8d "Parity metric").

    Returns a JSON-able dict: per DOF column m the metric E_m of the candidate against the reference, the reference's own
    fp64 noise floor N_m (its distance to the 80-bit solution `truth`), the distance `ref_vs_oracle` between the reference
    and the bit-faithful CPU restatement of its algorithm on the same inputs (what two correct fp64 implementations of the
    SAME algorithm differ by when only the LAPACK summation order changes), how many of the unknown columns meet the strict
    north-star bound E_m <= 1e-10, and the per-case relative error binned by the reference's own condition numbers
    (ExpertSolver(debug=True).conds(), expert.pyx:429-464).  `known_cols` are excluded from the counts (they must be
    bit-identical and are asserted separately)."""
    cand = np.asarray(cand, np.float64); ref = np.asarray(ref, np.float64)
    cols = [m for m in range(ref.shape[1]) if m not in set(known_cols)]
    E = column_metric(cand, ref)
    out = {"cases": int(ref.shape[0]), "columns": len(cols), "E": [float(E[m]) for m in cols], "E_max": float(max(E[m] for m in cols)),
           "strict_1e-10_columns": int(sum(E[m] <= TOL for m in cols))}
    if truth is not None:
        N = column_metric(ref, truth); T = column_metric(cand, truth)
        out["ref_noise_floor_N"] = [float(N[m]) for m in cols]
        out["cand_vs_truth"] = [float(T[m]) for m in cols]
        out["N_max"] = float(max(N[m] for m in cols))
        out["within_1e-10_plus_8N"] = bool(all(E[m] <= TOL + NOISE_MULT * N[m] for m in cols))
        # a column the reference itself resolves to 1e-11 must meet the strict bound: anything else is a bug
        out["resolved_columns_missing_strict"] = [int(m) for m in cols if N[m] < 1e-11 and E[m] > TOL]
    if oracle is not None:
        R = column_metric(np.asarray(oracle, np.float64), ref)
        out["ref_vs_oracle"] = [float(R[m]) for m in cols]
        out["ref_vs_oracle_max"] = float(max(R[m] for m in cols))
    if conds is not None:
        scale = np.nanmax(np.abs(ref), axis=0); scale = np.where(scale > 0, scale, 1.0)
        e_case = np.nanmax(np.abs(cand - ref)[:, cols] / scale[cols], axis=1)
        conds = np.asarray(conds, np.float64)
        edges = list(COND_EDGES) + [np.inf]
        hist = []
        for lo, hi in zip(edges[:-1], edges[1:]):
            sel = (conds >= lo) & (conds < hi)
            hist.append({"cond_lo": lo, "cond_hi": (hi if np.isfinite(hi) else None), "cases": int(sel.sum()),
                         "err_median": (float(np.median(e_case[sel])) if sel.any() else None),
                         "err_max": (float(e_case[sel].max()) if sel.any() else None)})
        out["err_vs_cond"] = hist
    return out
