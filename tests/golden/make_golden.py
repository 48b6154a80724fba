#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, cython, gcc, scipy).
It builds the reference's Cython sources where they lie (copied to a scratch dir
under /tmp, never into this repo), imports the resulting `wlsqm` package, and
records inputs + outputs (+ intermediates through a small Cython shim that
cimports the reference's own .pxd files).  Only the .npz vectors are committed;
the reference itself never travels.

    python3 tests/golden/make_golden.py [--scratch /tmp/wlsqm_oracle]
"""
import argparse
import os
import subprocess
import sys
import sysconfig

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import synth  # noqa: E402

REF = "/root/reference"
MODS = ["fitter/defs", "fitter/infra", "fitter/polyeval", "fitter/interp", "fitter/impl",
        "fitter/simple", "fitter/expert", "utils/ptrwrap", "utils/lapackdrivers"]

SHIM = r'''
# cython: wraparound=False, boundscheck=False, cdivision=True
# Golden-vector shim: reaches the reference's cdef-only internals through its own .pxd files.
from cython cimport view
cimport wlsqm.fitter.infra as infra
cimport wlsqm.fitter.impl as impl
import numpy as np

def remap(int n, long long mask):
    o2r = np.full((n,), -7, dtype=np.int32)
    r2o = np.full((n,), -7, dtype=np.int32)
    cdef int[::1] a = o2r
    cdef int[::1] b = r2o
    cdef int nr = infra.remap(&a[0], &b[0], n, mask)
    return nr, o2r, r2o

def intermediates(int dimension, int order, long long knowns, int wm, double[::1] xi, xk):
    """-> dict(o2r, r2o, c, w, A, row_scale, col_scale, LU, ipiv, nr, no)"""
    cdef double[::view.generic,::view.contiguous] xkManyD = None
    cdef double[::view.generic] xk1D = None
    cdef int nk = xk.shape[0]
    cdef double x0 = xi[0], x1 = 0., x2 = 0.
    if dimension >= 2:
        xkManyD = xk
        x1 = xi[1]
    else:
        xk1D = xk
    if dimension == 3:
        x2 = xi[2]
    cdef infra.Case* case = infra.Case_new(dimension, order, x0, x1, x2, nk, knowns, wm, 0, 0, <infra.CaseManager*>0, <infra.Case*>0)
    cdef int no = case.no, nr = case.nr, i
    impl.make_c_nD(case, xkManyD, xk1D)
    impl.make_A(case)
    out = dict(no=no, nr=nr)
    out["o2r"] = np.array([case.o2r[i] for i in range(no)], dtype=np.int32)
    out["r2o"] = np.array([case.r2o[i] for i in range(no)], dtype=np.int32)
    out["c"] = np.array([case.c[i] for i in range(nk*no)], dtype=np.float64).reshape(nk, no)
    out["w"] = np.array([case.w[i] for i in range(nk)], dtype=np.float64)
    nn = nr*nr if nr > 0 else 0
    out["A"] = np.array([case.A[i] for i in range(nn)], dtype=np.float64)
    impl.preprocess_A(case, 0)
    nr0 = nr if nr > 0 else 0
    out["row_scale"] = np.array([case.row_scale[i] for i in range(nr0)], dtype=np.float64)
    out["col_scale"] = np.array([case.col_scale[i] for i in range(nr0)], dtype=np.float64)
    out["LU"] = np.array([case.A[i] for i in range(nn)], dtype=np.float64)
    out["ipiv"] = np.array([case.ipiv[i] for i in range(nr0)], dtype=np.int32)
    infra.Case_del(case)
    return out
'''


def build_reference(scratch):
    """SURVEY.md §8(c) recipe: cython + gcc on the reference's own .pyx files, outputs in `scratch`."""
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    import scipy
    sp = os.path.dirname(scipy.__path__[0])
    inc = ["-I/usr/include/python3.10", "-I" + np.get_include(), "-Iwlsqm/fitter"]
    os.makedirs(scratch, exist_ok=True)
    if not os.path.isdir(os.path.join(scratch, "wlsqm")):
        subprocess.check_call(["cp", "-r", os.path.join(REF, "wlsqm"), scratch])
        subprocess.check_call(["chmod", "-R", "u+w", os.path.join(scratch, "wlsqm")])
    for m in MODS:
        so = os.path.join(scratch, "wlsqm", m + ext)
        if os.path.exists(so):
            continue
        subprocess.check_call(["cython", "-3", "-X", "language_level=3", "-I", ".", "-I", sp,
                               "wlsqm/%s.pyx" % m, "-o", "wlsqm/%s.c" % m], cwd=scratch)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-std=c11", "-D_USE_MATH_DEFINES", "-w"]
                              + inc + ["wlsqm/%s.c" % m, "-o", so, "-lm"], cwd=scratch)
    shim_so = os.path.join(scratch, "golden_shim" + ext)
    with open(os.path.join(scratch, "golden_shim.pyx"), "w") as f:
        f.write(SHIM)
    subprocess.check_call(["cython", "-3", "-I", ".", "-I", sp, "golden_shim.pyx", "-o", "golden_shim.c"], cwd=scratch)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-std=c11", "-w"] + inc
                          + ["golden_shim.c", "-o", shim_so, "-lm"], cwd=scratch)
    sys.path.insert(0, scratch)


# ----------------------------------------------------------------------------- case builders

def masks_for(no):
    """Sweep of knowns masks: none, F, single non-leading bit, multi-bit,
    all-but-one, all."""
    full = (1 << no) - 1
    out = [0, 1]
    if no >= 2:
        out.append(1 << (no - 1))                 # single non-leading bit (last DOF)
        out.append(full & ~(1 << (no // 2)))      # all-but-one
    if no >= 3:
        out.append(1 << (no // 2))                # single interior bit
        out.append((1 << 1) | (1 << (no - 1)))    # multi-bit, non-contiguous
    if no >= 6:
        out.append(1 | (1 << 2) | (1 << 4))       # multi-bit incl. F
    out.append(full)                              # everything known: nr = 0, no-op
    seen, res = set(), []
    for m in out:
        if m not in seen:
            seen.add(m); res.append(m)
    return res


def smooth(P):
    """A non-polynomial smooth test field on points P (n, dim)."""
    x = P[..., 0]
    f = np.sin(1.3 * x + 0.2) + 0.5 * x * x
    if P.shape[-1] >= 2:
        y = P[..., 1]
        f = f * np.cos(0.9 * y - 0.1) + 0.3 * np.exp(0.5 * y) * x
    if P.shape[-1] == 3:
        z = P[..., 2]
        f = f + np.sin(0.7 * z) * (1.0 + 0.4 * x * y)
    return f


def make_sweep(dim, rng):
    """Heterogeneous batch over orders x weighting x masks, ragged nk. Returns dict of arrays."""
    import wlsqm
    cases = []
    for order in range(5):
        no = wlsqm.number_of_dofs(dim, order)
        for wm in (wlsqm.WEIGHT_UNIFORM, wlsqm.WEIGHT_CENTER):
            for mask in masks_for(no):
                for rep in range(2):
                    nr = no - bin(mask).count("1")
                    nk = int(max(nr, 1) + 2 + rng.integers(0, min(2 * no + 3, 20)))
                    if rep == 1 and nr >= 1:
                        nk = max(nr, 1) + (0 if dim == 1 else 1)    # (nearly) determined stencil
                    cases.append((order, wm, mask, nk))
    n = len(cases)
    max_nk = max(c[3] for c in cases)
    max_no = wlsqm.number_of_dofs(dim, 4)
    xi = rng.uniform(-1.0, 1.0, (n, dim))
    h = rng.uniform(0.05, 0.4, (n, 1, 1))
    xk = xi[:, None, :] + h * rng.uniform(-1.0, 1.0, (n, max_nk, dim))
    fk = smooth(xk)
    fi_in = rng.uniform(-2.0, 2.0, (n, max_no))
    fi_in[:, 0] = smooth(xi)
    order = np.array([c[0] for c in cases], dtype=np.int32)
    wm = np.array([c[1] for c in cases], dtype=np.int32)
    knowns = np.array([c[2] for c in cases], dtype=np.int64)
    nk = np.array([c[3] for c in cases], dtype=np.int32)
    return dict(dim=dim, xk=xk, fk=fk, xi=xi, fi_in=fi_in, order=order, wm=wm, knowns=knowns, nk=nk)


def call_many(wlsqm, dim, variant, xk, fk, nk, xi, fi, sens, do_sens, order, knowns, wm, **kw):
    fn = getattr(wlsqm, "fit_%dD%s" % (dim, variant))
    if dim == 1:
        return fn(xk=np.ascontiguousarray(xk[..., 0]) if xk.ndim == 3 else xk, fk=fk, nk=nk,
                  xi=np.ascontiguousarray(xi[:, 0]) if xi.ndim == 2 else xi, fi=fi, sens=sens, do_sens=do_sens,
                  order=order, knowns=knowns, weighting_method=wm, **kw)
    return fn(xk=xk, fk=fk, nk=nk, xi=xi, fi=fi, sens=sens, do_sens=do_sens,
              order=order, knowns=knowns, weighting_method=wm, **kw)


def gen_sweep(dim, outdir):
    import wlsqm
    import golden_shim
    rng = np.random.default_rng(1000 + dim)
    d = make_sweep(dim, rng)
    n = len(d["nk"])
    max_nk = d["xk"].shape[1]
    max_no = d["fi_in"].shape[1]
    # basic, serial and parallel (must agree to 1e-14; we store the parallel one)
    fi = d["fi_in"].copy()
    call_many(wlsqm, dim, "_many_parallel", d["xk"], d["fk"], d["nk"], d["xi"], fi, None, 0,
              d["order"], d["knowns"], d["wm"], ntasks=4)
    fi_ser = d["fi_in"].copy()
    call_many(wlsqm, dim, "_many", d["xk"], d["fk"], d["nk"], d["xi"], fi_ser, None, 0,
              d["order"], d["knowns"], d["wm"])
    with np.errstate(invalid="ignore"):
        assert np.allclose(fi, fi_ser, rtol=1e-13, atol=1e-13, equal_nan=True)
    # sensitivities
    fi_s = d["fi_in"].copy()
    sens = np.full((n, max_nk, max_no), 777.0)
    call_many(wlsqm, dim, "_many", d["xk"], d["fk"], d["nk"], d["xi"], fi_s, sens, 1,
              d["order"], d["knowns"], d["wm"])
    # iterative
    fi_it = d["fi_in"].copy()
    iters = call_many(wlsqm, dim, "_iterative_many", d["xk"], d["fk"], d["nk"], d["xi"], fi_it, None, 0,
                      d["order"], d["knowns"], d["wm"], max_iter=10)
    # intermediates via the shim
    o2r = np.full((n, 35), -9, np.int32); r2o = np.full((n, 35), -9, np.int32)
    c = np.zeros((n, max_nk, 35)); w = np.zeros((n, max_nk))
    A = np.zeros((n, 35 * 35)); LU = np.zeros((n, 35 * 35))
    rs = np.zeros((n, 35)); cs = np.zeros((n, 35)); ipiv = np.zeros((n, 35), np.int32)
    for j in range(n):
        nkj = int(d["nk"][j])
        xkj = np.ascontiguousarray(d["xk"][j, :nkj, :] if dim > 1 else d["xk"][j, :nkj, 0])
        im = golden_shim.intermediates(dim, int(d["order"][j]), int(d["knowns"][j]), int(d["wm"][j]),
                                       np.ascontiguousarray(d["xi"][j], dtype=np.float64), xkj)
        no, nr = im["no"], max(im["nr"], 0)
        o2r[j, :no] = im["o2r"]; r2o[j, :no] = im["r2o"]
        c[j, :nkj, :no] = im["c"]; w[j, :nkj] = im["w"]
        A[j, :nr * nr] = im["A"]; LU[j, :nr * nr] = im["LU"]
        rs[j, :nr] = im["row_scale"]; cs[j, :nr] = im["col_scale"]; ipiv[j, :nr] = im["ipiv"]
    np.savez_compressed(os.path.join(outdir, "sweep_%dd.npz" % dim),
                        xk=d["xk"], fk=d["fk"], xi=d["xi"], fi_in=d["fi_in"], nk=d["nk"], order=d["order"],
                        knowns=d["knowns"], wm=d["wm"], fi=fi, fi_sens=fi_s, sens=sens, fi_iter=fi_it,
                        iters=np.int32(iters), o2r=o2r, r2o=r2o, c=c, w=w, A=A, LU=LU, row_scale=rs,
                        col_scale=cs, ipiv=ipiv)
    print("sweep_%dd: %d cases, max_nk=%d, iters=%d" % (dim, n, max_nk, iters))


def gen_remap(outdir):
    import golden_shim
    rows = []
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 4, 5, 6, 10, 15, 20, 35):
        masks = set(masks_for(n)) | {int(m) for m in rng.integers(0, 1 << n, 12)}
        masks |= {(1 << n), (1 << n) | 1} if n < 62 else set()      # bits >= n are not masked (infra.pyx:119)
        for m in sorted(masks):
            nr, o2r, r2o = golden_shim.remap(n, m)
            row = np.full(2 + 1 + 70, -9, np.int64)
            row[0], row[1], row[2] = n, m, nr
            row[3:3 + n] = o2r; row[38:38 + n] = r2o
            rows.append(row)
    np.savez_compressed(os.path.join(outdir, "remap.npz"), table=np.array(rows))
    print("remap: %d rows" % len(rows))


def gen_config(name, dim, order, nk, wm, knowns, npoints, ncases, outdir, extra_sens=16):
    """BASELINE.json configs at reduced case count: Halton + kNN (1D: sorted line), see synth.py."""
    import wlsqm
    if dim == 1:
        p = synth.line_problem_1d(npoints, nk // 2)
        for k in ("xk", "fk", "xi", "hoods"):
            p[k] = p[k][:ncases]
    else:
        p = synth.cloud_problem(dim, npoints, nk, ncases)
    no = wlsqm.number_of_dofs(dim, order)
    o = np.full(ncases, order, np.int32); kn = np.full(ncases, knowns, np.int64)
    w = np.full(ncases, wm, np.int32); nka = np.full(ncases, nk, np.int32)
    fi0 = np.zeros((ncases, no)); fi0[:, 0] = p["F"][:ncases]
    xk = p["xk"] if dim > 1 else p["xk"][..., None]
    xi = p["xi"] if dim > 1 else p["xi"][:, None]
    fi = fi0.copy()
    call_many(wlsqm, dim, "_many_parallel", xk, p["fk"], nka, xi, fi, None, 0, o, kn, w, ntasks=8)
    fi_it = fi0.copy()
    iters = call_many(wlsqm, dim, "_iterative_many_parallel", xk, p["fk"], nka, xi, fi_it, None, 0, o, kn, w,
                      max_iter=10, ntasks=8)
    ns = min(extra_sens, ncases)
    fi_s = fi0[:ns].copy(); sens = np.zeros((ns, nk, no))
    call_many(wlsqm, dim, "_many", xk[:ns], p["fk"][:ns], nka[:ns], xi[:ns], fi_s, sens, 1, o[:ns], kn[:ns], w[:ns])
    # ExpertSolver: prepare once, solve 3 time levels (C4 pattern), debug=True for conds
    xk_e = xk if dim > 1 else np.ascontiguousarray(xk[..., 0])
    xi_e = xi if dim > 1 else np.ascontiguousarray(xi[:, 0])
    solver = wlsqm.ExpertSolver(dimension=dim, nk=nka, order=o, knowns=kn, weighting_method=w,
                                algorithm=wlsqm.ALGO_BASIC, do_sens=False, ntasks=8, debug=True)
    solver.prepare(xi=xi_e, xk=xk_e)
    conds = solver.conds()
    fi_t = []
    for t in range(3):
        Ft = synth.field(p["S"], t=float(t)) if dim > 1 else p["F"] * (1.0 + 0.1 * t)
        fit = np.zeros((ncases, no)); fit[:, 0] = Ft[:ncases]
        solver.solve(fk=Ft[p["hoods"][:ncases]], fi=fit)
        fi_t.append(fit)
    assert np.array_equal(fi_t[0], fi) or np.allclose(fi_t[0], fi, rtol=1e-13, atol=1e-13)
    np.savez_compressed(os.path.join(outdir, "config_%s.npz" % name),
                        dim=dim, order=order, nk=nk, wm=wm, knowns=knowns, npoints=npoints, ncases=ncases,
                        hoods=p["hoods"][:ncases].astype(np.int32), fi=fi, fi_iter=fi_it, iters=np.int32(iters),
                        fi_sens=fi_s, sens=sens, conds=conds, fi_t=np.array(fi_t))
    print("config_%s: ncases=%d iters=%d cond_scaled median=%.3g max=%.3g" %
          (name, ncases, iters, np.median(conds), np.max(conds)))


def input_digest(*arrays):
    """sha256 over the bytes of the rebuilt inputs: lets a test assert that the arrays it rebuilds from synth.py are
    bit-identical to what the reference was given here."""
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def gen_config_dense(name, dim, order, nk, wm, knowns, npoints, outdir, every=977, ncases=1024):
    """BASELINE.json configs AT THE DENSITY THE METRIC IS QUOTED ON: the full `npoints`-point Halton cloud (1M for C2 / C3 /
    C5, 16M for configs[4]), every `every`-th point as a case (1024 cases: well under the reference's per-call limit,
    SURVEY.md section 6), neighbourhoods from a cKDTree over the WHOLE cloud.  Reference outputs: fit_*_many_parallel
    (simple.pyx:379-421), the iterative variant, ExpertSolver(debug=True).conds() (expert.pyx:429-464)."""
    import wlsqm
    S = synth.halton(npoints, dim)
    F = synth.field(S)
    cases = np.arange(ncases, dtype=np.int64) * every
    assert cases[-1] < npoints
    hoods = synth.knn(S, nk, query=cases).astype(np.int64)
    xk = S[hoods]; fk = F[hoods]; xi = S[cases].copy()
    # what a test rebuilds without the whole cloud must be the same bits
    xk2 = synth.halton_at(hoods, dim); xi2 = synth.halton_at(cases, dim)
    fk2 = synth.field(xk2.reshape(-1, dim)).reshape(ncases, nk); F2 = synth.field(xi2)
    assert np.array_equal(xk, xk2) and np.array_equal(xi, xi2) and np.array_equal(fk, fk2) and np.array_equal(F[cases], F2)
    no = wlsqm.number_of_dofs(dim, order)
    o = np.full(ncases, order, np.int32); kn = np.full(ncases, knowns, np.int64)
    w = np.full(ncases, wm, np.int32); nka = np.full(ncases, nk, np.int32)
    fi0 = np.zeros((ncases, no)); fi0[:, 0] = F[cases]
    fi = fi0.copy()
    call_many(wlsqm, dim, "_many_parallel", xk, fk, nka, xi, fi, None, 0, o, kn, w, ntasks=8)
    fi_it = fi0.copy()
    iters = call_many(wlsqm, dim, "_iterative_many_parallel", xk, fk, nka, xi, fi_it, None, 0, o, kn, w,
                      max_iter=10, ntasks=8)
    solver = wlsqm.ExpertSolver(dimension=dim, nk=nka, order=o, knowns=kn, weighting_method=w,
                                algorithm=wlsqm.ALGO_BASIC, do_sens=False, ntasks=8, debug=True)
    solver.prepare(xi=xi, xk=xk)
    conds = solver.conds()
    fi_e = fi0.copy()
    solver.solve(fk=fk, fi=fi_e)
    assert np.allclose(fi_e, fi, rtol=1e-13, atol=1e-13)
    np.savez_compressed(os.path.join(outdir, "config_%s.npz" % name),
                        dim=dim, order=order, nk=nk, wm=wm, knowns=knowns, npoints=npoints, ncases=ncases, every=every,
                        cases=cases, hoods=hoods.astype(np.int32), fi=fi, fi_iter=fi_it, iters=np.int32(iters), conds=conds,
                        digest=input_digest(xk, fk, xi, fi0))
    print("config_%s: %d cases of a %d-point cloud, iters=%d, cond_scaled median=%.3g max=%.3g" %
          (name, ncases, npoints, iters, np.median(conds), np.max(conds)))


DENSE = {   # name -> (dim, order, nk, wm, knowns, npoints)
    "C2_1M": (2, 2, 32, 2, 0, 1_000_000),
    "C3_1M": (2, 4, 64, 2, 1, 1_000_000),
    "C5_1M": (3, 2, 40, 2, 0, 1_000_000),
    "C5_16M": (3, 2, 40, 2, 0, 16_000_000),
}


def gen_c4_timelevels(outdir, npoints=1_000_000, every=977, ncases=1024, nlevels=4):
    """BASELINE.json configs[3] ("prepare once + 256 RHS solves") in the reference's own calling pattern (expert.pyx:309-426 prepare,
    :467-655 solve; tests/test_expert.py:92-117 re-solves one prepared geometry with new data): ONE ExpertSolver.prepare on the
    geometry of config_C2_1M (2D order 2, 32 neighbours, all DOFs unknown, every 977th case of the 1M-point cloud), then one
    solve() per time level t with fk_t = F_t[hoods], F_t = sin(pi x + 0.01 t) cos(pi y) (synth.field(S, t): SURVEY section 8d).
    The GPU's stacked solve (ExpertSolver.solve_many: the matrix-core kernel) is held to these outputs."""
    import wlsqm
    dim, order, nk = 2, 2, 32
    S = synth.halton(npoints, dim)
    cases = np.arange(ncases, dtype=np.int64) * every
    hoods = synth.knn(S, nk, query=cases).astype(np.int64)
    xk = S[hoods]; xi = S[cases].copy()
    assert np.array_equal(xk, synth.halton_at(hoods, dim)) and np.array_equal(xi, synth.halton_at(cases, dim))
    o = np.full(ncases, order, np.int32); kn = np.zeros(ncases, np.int64)
    w = np.full(ncases, wlsqm.WEIGHT_CENTER, np.int32); nka = np.full(ncases, nk, np.int32)
    solver = wlsqm.ExpertSolver(dimension=dim, nk=nka, order=o, knowns=kn, weighting_method=w,
                                algorithm=wlsqm.ALGO_BASIC, do_sens=False, ntasks=8, debug=False)
    solver.prepare(xi=xi, xk=xk)
    fis, fks = [], []
    for t in range(nlevels):
        Ft = synth.field(S, float(t))
        fk = Ft[hoods]
        assert np.array_equal(fk, synth.field(xk.reshape(-1, dim), float(t)).reshape(ncases, nk))
        fi = np.zeros((ncases, 6)); fi[:, 0] = Ft[cases]
        solver.solve(fk=fk, fi=fi)
        fis.append(fi); fks.append(fk)
    np.savez_compressed(os.path.join(outdir, "config_C4_1M.npz"), dim=dim, order=order, nk=nk, wm=int(wlsqm.WEIGHT_CENTER), knowns=0,
                        npoints=npoints, ncases=ncases, every=every, nlevels=nlevels, cases=cases, hoods=hoods.astype(np.int32),
                        fi=np.stack(fis), digest=input_digest(xk, np.stack(fks), xi))
    print("config_C4_1M: one prepare, %d time levels, %d cases of a %d-point cloud" % (nlevels, ncases, npoints))


def gen_edge(outdir):
    """Edge cases the reference's own tests pin (tests/test_edge_cases.py, test_stencil.py) + a few it does not."""
    import wlsqm
    out = {}
    rng = np.random.default_rng(42)
    # max_iter = 0 return value of the iterative driver (for/else quirk, impl.pyx:1016,1080-1081)
    xk = rng.uniform(-1, 1, (20, 2)); fk = np.sin(xk[:, 0]) * np.cos(xk[:, 1])
    for mi in (0, 1, 2, 10):
        fi = np.zeros(6)
        it = wlsqm.fit_2D_iterative(xk=xk, fk=fk, xi=np.zeros(2), fi=fi, sens=None, do_sens=0, order=2,
                                    knowns=0, weighting_method=wlsqm.WEIGHT_CENTER, max_iter=mi)
        out["iter_mi%d_fi" % mi] = fi; out["iter_mi%d_it" % mi] = np.int32(it)
    out["iter_xk"] = xk; out["iter_fk"] = fk
    # 5-point stencil with knowns = b2_XY (tests/test_stencil.py:134): known stays bit-identical
    h = 0.1
    st = np.array([[h, 0], [-h, 0], [0, h], [0, -h], [0, 0.0]])
    fst = np.exp(st[:, 0]) * np.sin(st[:, 1] + 0.3)
    fi = np.zeros(6); fi[4] = 0.0
    wlsqm.fit_2D(xk=st, fk=fst, xi=np.zeros(2), fi=fi, sens=None, do_sens=0, order=2,
                 knowns=wlsqm.b2_XY, weighting_method=wlsqm.WEIGHT_UNIFORM)
    out["stencil_xk"] = st; out["stencil_fk"] = fst; out["stencil_fi"] = fi
    # strided (non-contiguous outer axes) inputs through the _many API
    n, nk = 12, 14
    big_xk = rng.uniform(-1, 1, (2 * n, 2 * nk, 2)); big_fk = rng.uniform(-1, 1, (2 * n, 2 * nk))
    xkv = big_xk[::2, ::2, :]; fkv = big_fk[::2, ::2]
    xi = rng.uniform(-0.1, 0.1, (2 * n, 2))[::2]
    big_fi = np.zeros((2 * n, 8)); fiv = big_fi[::2, :6]
    o = np.full(2 * n, 2, np.int32)[::2]; kn = np.zeros(2 * n, np.int64)[::2]
    w = np.full(2 * n, 2, np.int32)[::2]; nka = np.full(2 * n, nk, np.int32)[::2]
    wlsqm.fit_2D_many(xk=xkv, fk=fkv, nk=nka, xi=xi, fi=fiv, sens=None, do_sens=0, order=o, knowns=kn,
                      weighting_method=w)
    out["strided_big_xk"] = big_xk; out["strided_big_fk"] = big_fk; out["strided_xi"] = np.ascontiguousarray(xi)
    out["strided_fi"] = np.ascontiguousarray(fiv); out["strided_big_fi_after"] = big_fi
    np.savez_compressed(os.path.join(outdir, "edge.npz"), **out)
    print("edge: %d arrays" % len(out))


def gen_testmany2d(outdir):
    """The harness pattern of examples/wlsqm_example.py:55-187 (testmany2d): neighbourhoods from a radius query
    (ragged nk, capped at max_nk), order 4, knowns = b2_F, WEIGHT_CENTER; driver call and ExpertSolver."""
    import wlsqm
    from scipy.spatial import cKDTree
    N, r, max_nk = 1024, 0.14, 100
    S = synth.halton(N, 2)
    F = synth.field(S)
    tree = cKDTree(S)
    lists = tree.query_ball_point(S, r)
    hoods = np.full((N, max_nk), -1, np.int32); nk = np.zeros(N, np.int32)
    for i, L in enumerate(lists):
        L = [j for j in L if j != i][:max_nk]
        nk[i] = len(L); hoods[i, :len(L)] = L
    assert nk.min() >= 15, nk.min()
    hp = np.where(hoods >= 0, hoods, 0)
    xk = S[hp]; fk = F[hp]
    o = np.full(N, 4, np.int32); kn = np.full(N, wlsqm.b2_F, np.int64); w = np.full(N, wlsqm.WEIGHT_CENTER, np.int32)
    fi = np.zeros((N, 15)); fi[:, 0] = F
    wlsqm.fit_2D_many_parallel(xk=xk, fk=fk, nk=nk, xi=S, fi=fi, sens=None, do_sens=0, order=o, knowns=kn,
                               weighting_method=w, ntasks=8)
    s = wlsqm.ExpertSolver(dimension=2, nk=nk, order=o, knowns=kn, weighting_method=w, ntasks=8)
    s.prepare(xi=S, xk=xk)
    fi2 = np.zeros((N, 15)); fi2[:, 0] = F
    s.solve(fk=fk, fi=fi2)
    assert np.allclose(fi, fi2, rtol=1e-13, atol=1e-13)
    np.savez_compressed(os.path.join(outdir, "testmany2d.npz"), hoods=hoods, nk=nk, fi=fi, N=N, r=r)
    print("testmany2d: nk in [%d, %d]" % (nk.min(), nk.max()))


def gen_interp(outdir):
    """interpolate_fit (interp.pyx:34-143) for every (dimension, order, diff), and ExpertSolver.interpolate
    (expert.pyx:687-781) in both modes on a small 2D cloud."""
    import wlsqm
    rng = np.random.default_rng(11)
    out = {}
    for dim in (1, 2, 3):
        for order in range(5):
            no = wlsqm.number_of_dofs(dim, order)
            fi = rng.uniform(-1, 1, no)
            xi = rng.uniform(-0.5, 0.5, dim)
            x = xi + rng.uniform(-0.3, 0.3, (17, dim))
            vals = np.zeros((no + 2, 17))
            for diff in range(no + 2):                      # two past-the-end values: must give zeros (interp.pyx:674-678)
                if dim == 1:
                    vals[diff] = wlsqm.interpolate_fit(float(xi[0]), fi, dim, order, np.ascontiguousarray(x[:, 0]), diff)
                else:
                    vals[diff] = wlsqm.interpolate_fit(xi, fi, dim, order, x, diff)
            k = "d%do%d_" % (dim, order)
            out[k + "fi"] = fi; out[k + "xi"] = xi; out[k + "x"] = x; out[k + "vals"] = vals
    # ExpertSolver.interpolate on a 2D order-2 cloud
    p = synth.cloud_problem(2, 2048, 16, 300)
    n = 300
    nk = np.full(n, 16, np.int32); o = np.full(n, 2, np.int32); o[::3] = 3
    kn = np.zeros(n, np.int64); w = np.full(n, 2, np.int32)
    solver = wlsqm.ExpertSolver(dimension=2, nk=nk, order=o, knowns=kn, weighting_method=w, algorithm=wlsqm.ALGO_BASIC)
    solver.prepare(xi=p["xi"], xk=p["xk"])
    fi = np.zeros((n, 10))
    solver.solve(fk=p["fk"], fi=fi)
    solver.prep_interpolate()
    xq = p["xi"][rng.integers(0, n, 64)] + rng.uniform(-0.01, 0.01, (64, 2))
    for diff in (0, 1, 4, 7):
        v, I = solver.interpolate(xq, mode="nearest", diff=diff)
        out["ex_nearest_%d" % diff] = v; out["ex_I"] = np.asarray(I, np.int64)
        v2, _ = solver.interpolate(xq, mode="continuous", r=0.05, diff=diff)
        out["ex_cont_%d" % diff] = v2
    out["ex_xq"] = xq; out["ex_order"] = o; out["ex_fi"] = fi
    np.savez_compressed(os.path.join(outdir, "interp.npz"), **out)
    print("interp: %d arrays" % len(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, help="generate just one fixture family (e.g. interp)")
    ap.add_argument("--scratch", default="/tmp/wlsqm_oracle")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    build_reference(a.scratch)
    import wlsqm
    assert os.path.realpath(wlsqm.__file__).startswith(os.path.realpath(a.scratch)), wlsqm.__file__
    if a.only == "interp":
        gen_interp(a.out)
        return
    if a.only == "testmany2d":
        gen_testmany2d(a.out)
        return
    if a.only == "c4":
        gen_c4_timelevels(a.out)
        return
    if a.only == "dense":
        for name, (dim, order, nk, wm, kn, npts) in DENSE.items():
            gen_config_dense(name, dim, order, nk, wm, kn, npts, a.out)
        return
    gen_remap(a.out)
    for dim in (1, 2, 3):
        gen_sweep(dim, a.out)
    gen_edge(a.out)
    gen_config("C1", 1, 2, 8, wlsqm.WEIGHT_UNIFORM, 0, 10000, 512, a.out)
    gen_config("C2", 2, 2, 32, wlsqm.WEIGHT_CENTER, 0, 16384, 512, a.out)
    gen_config("C3", 2, 4, 64, wlsqm.WEIGHT_CENTER, wlsqm.b2_F, 16384, 512, a.out)
    gen_config("C5", 3, 2, 40, wlsqm.WEIGHT_CENTER, 0, 32768, 512, a.out)
    gen_config("X3", 3, 4, 100, wlsqm.WEIGHT_CENTER, wlsqm.b3_F, 32768, 64, a.out, extra_sens=4)
    gen_config("X2", 2, 3, 40, wlsqm.WEIGHT_CENTER, wlsqm.b2_F, 16384, 256, a.out)
    gen_interp(a.out)
    gen_testmany2d(a.out)
    for name, (dim, order, nk, wm, kn, npts) in DENSE.items():
        gen_config_dense(name, dim, order, nk, wm, kn, npts, a.out)
    gen_c4_timelevels(a.out)


if __name__ == "__main__":
    main()
