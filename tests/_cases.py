"""Builders for test inputs from the committed golden fixtures (tests/golden/*.npz) and synth.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-wlsqm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
NDOF = {1: [1, 2, 3, 4, 5], 2: [1, 3, 6, 10, 15], 3: [1, 4, 10, 20, 35]}
CONFIGS = ("C1", "C2", "C3", "C5", "X2", "X3")
DENSE = ("C2_1M", "C3_1M", "C5_1M", "C5_16M")     # every 977th case of the FULL-density clouds the metric is quoted on


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def sweep(dim):
    """Heterogeneous sweep batch; 1D arrays are squeezed to the reference's 1D layout."""
    g = golden("sweep_%dd.npz" % dim)
    d = {k: g[k] for k in g.files}
    if dim == 1:
        d["xk"] = np.ascontiguousarray(d["xk"][..., 0])
        d["xi"] = np.ascontiguousarray(d["xi"][:, 0])
    return d


def config(name):
    """Rebuild the dense inputs of a BASELINE-style config fixture (Halton + stored kNN indices)."""
    g = golden("config_%s.npz" % name)
    dim, order, nk, n = int(g["dim"]), int(g["order"]), int(g["nk"]), int(g["ncases"])
    if dim == 1:
        p = synth.line_problem_1d(int(g["npoints"]), nk // 2)
        S, F = p["S"], p["F"]
    else:
        S = synth.halton(int(g["npoints"]), dim)
        F = synth.field(S)
    hoods = g["hoods"].astype(np.int64)
    no = NDOF[dim][order]
    fi0 = np.zeros((n, no)); fi0[:, 0] = F[:n]
    d = dict(g=g, dim=dim, order=order, nkv=nk, n=n, no=no, S=S, F=F, hoods=hoods,
             xk=S[hoods], fk=F[hoods], xi=S[:n].copy(), fi0=fi0,
             order_a=np.full(n, order, np.int32), knowns_a=np.full(n, int(g["knowns"]), np.int64),
             wm_a=np.full(n, int(g["wm"]), np.int32), nk_a=np.full(n, nk, np.int32))
    return d


def scaled_cond(g, j, no, knowns):
    """2-norm condition number of the reference's Ruiz-scaled matrix of sweep case j (from the golden A, scales)."""
    nr = no - bin(int(knowns) & ((1 << no) - 1)).count("1") - bin(int(knowns) >> no).count("1")
    if nr < 1:
        return 1.0
    A = g["A"][j, :nr * nr].reshape(nr, nr, order="F")
    As = A * g["row_scale"][j, :nr, None] * g["col_scale"][j, None, :nr]
    return float(np.linalg.cond(As))


def config_dense(name):
    """Dense inputs of a full-density fixture (tests/golden/make_golden.py gen_config_dense): 1024 cases picked from the 1M /
    16M-point Halton cloud, rebuilt point by point (synth.halton_at) and checked bit-for-bit against the digest of what the
    reference was given."""
    import hashlib
    g = golden("config_%s.npz" % name)
    dim, order, nk, n = int(g["dim"]), int(g["order"]), int(g["nk"]), int(g["ncases"])
    hoods, cases = g["hoods"].astype(np.int64), g["cases"]
    xk = synth.halton_at(hoods, dim); xi = synth.halton_at(cases, dim)
    fk = synth.field(xk.reshape(-1, dim)).reshape(n, nk)
    no = NDOF[dim][order]
    fi0 = np.zeros((n, no)); fi0[:, 0] = synth.field(xi)
    h = hashlib.sha256()
    for a in (xk, fk, xi, fi0):
        h.update(np.ascontiguousarray(a).tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), g["digest"]), \
        "rebuilt inputs of %s differ from what the reference was given (numpy / libm difference on this host?)" % name
    return dict(g=g, dim=dim, order=order, nkv=nk, n=n, no=no, xk=xk, fk=fk, xi=xi, fi0=fi0, conds=g["conds"],
                order_a=np.full(n, order, np.int32), knowns_a=np.full(n, int(g["knowns"]), np.int64),
                wm_a=np.full(n, int(g["wm"]), np.int32), nk_a=np.full(n, nk, np.int32))


def config_c4():
    """BASELINE configs[3] in the reference's calling pattern (tests/golden/config_C4_1M.npz: one ExpertSolver.prepare on the
    geometry of config_C2_1M, one solve() per time level with fk_t = F_t[hoods]): the rebuilt geometry, the stacked fields
    fk (nlevels, n, nk), the start values fi0 (nlevels, n, 6) and the reference's outputs."""
    import hashlib
    g = golden("config_C4_1M.npz")
    dim, nk, n, L = int(g["dim"]), int(g["nk"]), int(g["ncases"]), int(g["nlevels"])
    hoods, cases = g["hoods"].astype(np.int64), g["cases"]
    xk = synth.halton_at(hoods, dim); xi = synth.halton_at(cases, dim)
    fk = np.stack([synth.field(xk.reshape(-1, dim), float(t)).reshape(n, nk) for t in range(L)])
    fi0 = np.zeros((L, n, 6))
    for t in range(L):
        fi0[t, :, 0] = synth.field(xi, float(t))
    h = hashlib.sha256()
    for a in (xk, fk, xi):
        h.update(np.ascontiguousarray(a).tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), g["digest"]), "rebuilt inputs of config_C4_1M differ"
    return dict(g=g, dim=dim, order=int(g["order"]), nkv=nk, n=n, nlevels=L, xk=xk, xi=xi, fk=fk, fi0=fi0, fi_ref=g["fi"],
                order_a=np.full(n, int(g["order"]), np.int32), knowns_a=np.zeros(n, np.int64),
                wm_a=np.full(n, int(g["wm"]), np.int32), nk_a=np.full(n, nk, np.int32))
