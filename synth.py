"""Deterministic synthetic inputs for the WLSQM hot path.

Shared by bench.py, tests/ and tests/golden/make_golden.py.  Pure numpy/scipy,
no RNG for the Halton clouds, so the same inputs are rebuilt on any box.

Points: first N points of the unscrambled Halton sequence in [0,1]^dim (bases
2,3,5), skipping index 0.  Neighbourhoods: nk nearest neighbours excluding
self (cKDTree).  Field: sin(pi x) cos(pi y) [exp(z)].  This mirrors the calling
convention of the reference's examples/expertsolver_example.py:48-92
(kNN query with 1+nk, drop self, xk = S[hoods], fk = F[hoods]).
"""
import numpy as np

_PRIMES = (2, 3, 5)


def halton(n, dim, skip=1):
    """Unscrambled Halton points, shape (n, dim), C-contiguous float64."""
    idx = np.arange(skip, skip + n, dtype=np.int64)
    out = np.empty((n, dim), dtype=np.float64)
    for d in range(dim):
        b = _PRIMES[d]
        i = idx.copy()
        f = 1.0
        r = np.zeros(n, dtype=np.float64)
        while np.any(i > 0):
            f = f / b
            r += f * (i % b)
            i //= b
        out[:, d] = r
    return out


def halton_at(index, dim, skip=1):
    """Rows `index` (any integer array) of halton(n, dim, skip) without building the whole cloud: bit-identical to
    halton(n, dim, skip)[index] (same operations in the same order per point)."""
    index = np.asarray(index, dtype=np.int64)
    idx = index.ravel() + skip
    out = np.empty((idx.size, dim), dtype=np.float64)
    for d in range(dim):
        b = _PRIMES[d]
        i = idx.copy()
        f = 1.0
        r = np.zeros(idx.size, dtype=np.float64)
        while np.any(i > 0):
            f = f / b
            r += f * (i % b)
            i //= b
        out[:, d] = r
    return out.reshape(index.shape + (dim,))


def field(S, t=0.0):
    """sin(pi x + 0.01 t) cos(pi y) [exp(z)] on points S (n, dim); 1D: sin(2 pi x)."""
    S = np.asarray(S)
    if S.ndim == 1:
        return np.sin(2.0 * np.pi * S)
    x = S[:, 0]
    if S.shape[1] == 1:
        return np.sin(2.0 * np.pi * x)
    f = np.sin(np.pi * x + 0.01 * t) * np.cos(np.pi * S[:, 1])
    if S.shape[1] == 3:
        f = f * np.exp(S[:, 2])
    return f


def knn(S, nk, query=None, workers=-1):
    """Indices (nq, nk) int32 of the nk nearest neighbours of each query point, self excluded."""
    from scipy.spatial import cKDTree
    tree = cKDTree(S)
    if query is None:
        _, idx = tree.query(S, nk + 1, workers=workers)
        return np.ascontiguousarray(idx[:, 1:]).astype(np.int32)
    _, idx = tree.query(S[query], nk + 1, workers=workers)
    return np.ascontiguousarray(idx[:, 1:]).astype(np.int32)


def cloud_problem(dim, npoints, nk, ncases=None, hoods=None):
    """Dense reference-API arrays for `ncases` fits on a Halton cloud of `npoints`.

    Returns dict(S, F, hoods, xk, fk, xi) with xk (ncases, nk, dim), fk (ncases, nk),
    xi (ncases, dim).  If `hoods` is given it is used instead of a kNN search.
    """
    S = halton(npoints, dim)
    F = field(S)
    if ncases is None:
        ncases = npoints
    if hoods is None:
        hoods = knn(S, nk, query=np.arange(ncases))
    hoods = np.asarray(hoods, dtype=np.int64)
    return dict(S=S, F=F, hoods=hoods.astype(np.int32), xk=S[hoods], fk=F[hoods], xi=S[:ncases].copy())


def line_problem_1d(npoints=10000, half=4, seed=0):
    """C1: sorted uniform points on [0,1], `half` neighbours each side
    (windows shifted inward at the ends), F = sin(2 pi x).  Returns dict like cloud_problem."""
    x = np.sort(np.random.default_rng(seed).uniform(0.0, 1.0, npoints))
    F = np.sin(2.0 * np.pi * x)
    nk = 2 * half
    i = np.arange(npoints)
    start = np.clip(i - half, 0, npoints - (nk + 1))
    win = start[:, None] + np.arange(nk + 1)[None, :]          # nk+1 consecutive points incl. self
    mask = win != i[:, None]
    hoods = win[mask].reshape(npoints, nk)
    return dict(S=x, F=F, hoods=hoods.astype(np.int32), xk=x[hoods], fk=F[hoods], xi=x.copy())


def morton_order(S, bits=16):
    """Permutation that sorts points of [0,1]^dim along a Z-order (Morton) curve, so that points close in
    space are close in memory — what the index-based kernels want for L2 locality of the neighbour gathers."""
    S = np.asarray(S)
    if S.ndim == 1:
        return np.argsort(S, kind="stable")
    dim = S.shape[1]
    q = np.clip((S * (1 << bits)).astype(np.uint64), 0, (1 << bits) - 1)
    key = np.zeros(S.shape[0], dtype=np.uint64)
    for b in range(bits):
        for d in range(dim):
            key |= ((q[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * dim + d)
    return np.argsort(key, kind="stable")
