"""Simple ("LAPACK driver style") fitting API, backed by the MI355X HIP kernels.

Drop-in mirror of the reference's wlsqm/fitter/simple.pyx Python API (simple.pyx:60-604):
same function names, argument names, defaults, array typing rules, in/out semantics of
``fi`` (knowns read, unknowns written, knowns left bit-identical) and return values
(0 for the basic algorithm, max refinement iterations for the iterative one).

``ntasks`` (OpenMP thread count in the reference) and ``debug`` are accepted and ignored:
the batch runs on one GPU as a single launch per polynomial order.
"""
import ctypes as C

import numpy as np

from . import defs
from .. import _binding as B

__all__ = [
    "fit_1D", "fit_1D_iterative", "fit_1D_many", "fit_1D_iterative_many",
    "fit_1D_many_parallel", "fit_1D_iterative_many_parallel",
    "fit_2D", "fit_2D_iterative", "fit_2D_many", "fit_2D_iterative_many",
    "fit_2D_many_parallel", "fit_2D_iterative_many_parallel",
    "fit_3D", "fit_3D_iterative", "fit_3D_many", "fit_3D_iterative_many",
    "fit_3D_many_parallel", "fit_3D_iterative_many_parallel",
]


def _run_many(dimension, xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method,
              iterative, max_iter, ntasks=1):
    """Common driver: replaces generic_fit_basic_many[_parallel] / generic_fit_iterative_many[_parallel]
    (simple.pyx:731-831, 850-942, 953-1058, 1065-1170)."""
    if ntasks is None or int(ntasks) < 1:
        raise ValueError("ntasks must be >= 1, got %r" % (ntasks,))
    nk = B.view(nk, np.int32, 1, "nk")
    order = B.view(order, np.int32, 1, "order")
    knowns = B.view(knowns, np.int64, 1, "knowns")
    wm = B.view(weighting_method, np.int32, 1, "weighting_method")
    fk = B.view(fk, np.float64, 2, "fk")
    fi = B.view(fi, np.float64, 2, "fi", contiguous_last=True, writable=True)
    if dimension == 1:
        xk = B.view(xk, np.float64, 2, "xk")
        xi = B.view(xi, np.float64, 1, "xi")
    else:
        xk = B.view(xk, np.float64, 3, "xk", contiguous_last=True)
        xi = B.view(xi, np.float64, 2, "xi", contiguous_last=True)
        if xk.shape[2] < dimension or xi.shape[1] < dimension:
            raise ValueError("xk/xi must have %d coordinates on the last axis" % dimension)
    ncases = nk.shape[0]
    for name, a in (("order", order), ("knowns", knowns), ("weighting_method", wm), ("fk", fk), ("fi", fi),
                    ("xk", xk), ("xi", xi)):
        if a.shape[0] < ncases:
            raise ValueError("%s has fewer entries (%d) than nk (%d)" % (name, a.shape[0], ncases))
    if ncases < 1:
        raise ValueError("max_cases must be >= 1, got %d" % ncases)          # infra.pyx:311-313
    if np.any((order < 0) | (order > 4)):
        raise ValueError("order must be 0, 1, 2, 3 or 4")
    lib = B.lib()
    max_no = lib.wlsqm_hip_number_of_dofs(dimension, int(order.max()))       # no grows with the order
    max_nk = int(nk.max())
    if max_nk > min(xk.shape[1], fk.shape[1]):
        raise ValueError("max(nk) = %d exceeds the neighbour axis of xk/fk" % max_nk)
    if fi.shape[1] < max_no:
        raise ValueError("fi has %d columns, need at least %d" % (fi.shape[1], max_no))
    do_sens = 1 if do_sens else 0
    if do_sens:
        if sens is None:
            raise ValueError("do_sens is set but sens is None")
        sens = B.view(sens, np.float64, 3, "sens", contiguous_last=True, writable=True)
        if sens.shape[0] < ncases or sens.shape[1] < max_nk or sens.shape[2] < max_no:
            raise ValueError("sens is too small: need at least (%d, %d, %d)" % (ncases, max_nk, max_no))
    else:
        sens = None

    b = B.Batch()
    b.dimension, b.do_sens, b.ncases = dimension, do_sens, ncases
    b.xk, b.xk_stride_case, b.xk_stride_k = xk.ctypes.data, B.es(xk, 0), B.es(xk, 1)
    b.fk, b.fk_stride_case, b.fk_stride_k = fk.ctypes.data, B.es(fk, 0), B.es(fk, 1)
    b.nk, b.nk_stride = nk.ctypes.data, B.es(nk, 0)
    b.xi, b.xi_stride_case = xi.ctypes.data, B.es(xi, 0)
    b.fi, b.fi_stride_case = fi.ctypes.data, B.es(fi, 0)
    if sens is not None:
        b.sens, b.sens_stride_case, b.sens_stride_k = sens.ctypes.data, B.es(sens, 0), B.es(sens, 1)
    b.order, b.order_stride = order.ctypes.data, B.es(order, 0)
    b.knowns, b.knowns_stride = knowns.ctypes.data, B.es(knowns, 0)
    b.weighting_method, b.wm_stride = wm.ctypes.data, B.es(wm, 0)
    b.iterative, b.max_iter, b.max_nk = (1 if iterative else 0), int(max_iter), max_nk
    its = C.c_int32(0)
    B.check(lib.wlsqm_hip_fit_many_host(C.byref(b), B.default_device(), C.byref(its)))
    return int(its.value)


def _run_one(dimension, xk, fk, xi, fi, sens, do_sens, order, knowns, weighting_method, iterative, max_iter):
    """Single case: replaces generic_fit_basic / generic_fit_iterative (simple.pyx:620-696)."""
    fk = B.view(fk, np.float64, 1, "fk")
    fi = B.view(fi, np.float64, 1, "fi", contiguous_last=True, writable=True)
    nkv = fk.shape[0]                                                       # simple.pyx:637
    if dimension == 1:
        xk = B.view(xk, np.float64, 1, "xk")
        xk_b = xk[None, :]
        xi_b = np.array([float(xi)], dtype=np.float64)
    else:
        xk = B.view(xk, np.float64, 2, "xk", contiguous_last=True)
        xi = B.view(xi, np.float64, 1, "xi", contiguous_last=True)
        xk_b = xk[None, :, :]
        xi_b = xi[None, :]
    sens_b = None
    if do_sens:
        if sens is None:
            raise ValueError("do_sens is set but sens is None")
        sens_b = B.view(sens, np.float64, 2, "sens", contiguous_last=True, writable=True)[None, :, :]
    return _run_many(dimension, xk_b, fk[None, :], np.array([nkv], np.int32), xi_b, fi[None, :], sens_b, do_sens,
                     np.array([order], np.int32), np.array([knowns], np.int64),
                     np.array([weighting_method], np.int32), iterative, max_iter)


def _make(dimension):
    bF = {1: defs.b1_F, 2: defs.b2_F, 3: defs.b3_F}[dimension]
    D = "%dD" % dimension

    def fit(xk, fk, xi, fi, sens, do_sens=0, order=2, knowns=bF, weighting_method=defs.WEIGHT_CENTER, debug=0):
        return _run_one(dimension, xk, fk, xi, fi, sens, do_sens, order, knowns, weighting_method, False, 0)

    def fit_iterative(xk, fk, xi, fi, sens, do_sens=0, order=2, knowns=bF, weighting_method=defs.WEIGHT_CENTER,
                      max_iter=10, debug=0):
        return _run_one(dimension, xk, fk, xi, fi, sens, do_sens, order, knowns, weighting_method, True, max_iter)

    def fit_many(xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method, debug=0):
        return _run_many(dimension, xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method, False, 0)

    def fit_iterative_many(xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method, max_iter=10, debug=0):
        return _run_many(dimension, xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method, True, max_iter)

    def fit_many_parallel(xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method, ntasks=8, debug=0):
        return _run_many(dimension, xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method, False, 0,
                         ntasks=ntasks)

    def fit_iterative_many_parallel(xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method,
                                    max_iter=10, ntasks=8, debug=0):
        return _run_many(dimension, xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method, True,
                         max_iter, ntasks=ntasks)

    ref = {1: ("429-478", "481-498", "501-537", "540-559", "562-581", "584-604"),
           2: ("241-290", "293-315", "318-354", "357-376", "379-398", "401-421"),
           3: ("60-109", "111-128", "131-167", "170-189", "192-211", "214-234")}[dimension]
    out = {}
    for f, suffix, lines, what in (
            (fit, "", ref[0], "Fit one local model"),
            (fit_iterative, "_iterative", ref[1], "Fit one local model, with iterative refinement"),
            (fit_many, "_many", ref[2], "Fit many local models"),
            (fit_iterative_many, "_iterative_many", ref[3], "Fit many local models, with iterative refinement"),
            (fit_many_parallel, "_many_parallel", ref[4], "Fit many local models (ntasks accepted, ignored on GPU)"),
            (fit_iterative_many_parallel, "_iterative_many_parallel", ref[5],
             "Fit many local models, with iterative refinement (ntasks accepted, ignored on GPU)")):
        name = "fit_%s%s" % (D, suffix)
        f.__name__ = f.__qualname__ = name
        f.__doc__ = ("%s to %s scalar data on one MI355X.\n\nSame arguments, in/out semantics and return value as "
                     "the reference's wlsqm.fitter.simple.%s (simple.pyx:%s).\nAll arrays are allocated by the "
                     "caller; dtype float64 (xk, fk, xi, fi, sens), int32 (nk, order, weighting_method), int64 (knowns)."
                     % (what, D, name, lines))
        out[name] = f
    return out


globals().update(_make(1))
globals().update(_make(2))
globals().update(_make(3))
