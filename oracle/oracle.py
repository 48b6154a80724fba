"""ctypes binding of oracle/libwlsqm_oracle.so — TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(see the header of wlsqm_oracle.c).  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleDebug(C.Structure):
    _fields_ = [("o2r", C.c_void_p), ("r2o", C.c_void_p), ("c", C.c_void_p), ("w", C.c_void_p),
                ("A", C.c_void_p), ("row_scale", C.c_void_p), ("col_scale", C.c_void_p),
                ("LU", C.c_void_p), ("ipiv", C.c_void_p), ("max_nk", C.c_long)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libwlsqm_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.wlsqm_oracle_number_of_dofs.argtypes = [C.c_int, C.c_int]
        L.wlsqm_oracle_number_of_reduced_dofs.argtypes = [C.c_int, C.c_longlong]
        L.wlsqm_oracle_remap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong]
        L.wlsqm_oracle_fit_many.restype = C.c_int
        L.wlsqm_oracle_fit_many.argtypes = (
            [C.c_int, C.c_long] + [C.c_void_p, C.c_long, C.c_long] * 2 + [C.c_void_p, C.c_long] * 3
            + [C.c_void_p, C.c_long, C.c_long, C.c_int] + [C.c_void_p, C.c_long] * 3
            + [C.c_int, C.c_int, C.c_int, C.c_void_p])
        _LIB = L
    return _LIB


def number_of_dofs(dimension, order):
    return lib().wlsqm_oracle_number_of_dofs(dimension, order)


def remap(n, mask):
    o2r = np.full(n, -7, np.int32)
    r2o = np.full(n, -7, np.int32)
    nr = lib().wlsqm_oracle_remap(o2r.ctypes.data, r2o.ctypes.data, n, mask)
    return nr, o2r, r2o


def _es(a, axis):
    return a.strides[axis] // a.itemsize


def fit_many(dimension, xk, fk, nk, xi, fi, sens, do_sens, order, knowns, weighting_method,
             iterative=False, max_iter=10, ntasks=1, debug_capture=False):
    """Same argument meaning as the reference's fit_*D_many (simple.pyx:131-167 etc.); strided
    views are honoured.  Returns iterations taken (0 for the basic algorithm), and the captured
    intermediates dict if debug_capture."""
    ncases = nk.shape[0]
    assert xk.dtype == np.float64 and fk.dtype == np.float64 and fi.dtype == np.float64
    assert nk.dtype == np.int32 and order.dtype == np.int32 and weighting_method.dtype == np.int32
    assert knowns.dtype == np.int64
    if dimension == 1:
        sx_j, sx_k = _es(xk, 0), _es(xk, 1)
        assert xi.ndim == 1
        sxi = _es(xi, 0)
    else:
        assert xk.strides[2] == 8 and xi.strides[1] == 8
        sx_j, sx_k = _es(xk, 0), _es(xk, 1)
        sxi = _es(xi, 0)
    assert fi.strides[1] == 8
    dbg = None
    cap = None
    if debug_capture:
        max_nk = xk.shape[1]
        cap = dict(o2r=np.full((ncases, 35), -9, np.int32), r2o=np.full((ncases, 35), -9, np.int32),
                   c=np.zeros((ncases, max_nk, 35)), w=np.zeros((ncases, max_nk)),
                   A=np.zeros((ncases, 1225)), row_scale=np.zeros((ncases, 35)),
                   col_scale=np.zeros((ncases, 35)), LU=np.zeros((ncases, 1225)),
                   ipiv=np.zeros((ncases, 35), np.int32))
        dbg = OracleDebug(*[cap[k].ctypes.data for k in
                            ("o2r", "r2o", "c", "w", "A", "row_scale", "col_scale", "LU", "ipiv")], max_nk)
    if sens is not None:
        assert sens.strides[2] == 8
        sp, ss_j, ss_k = sens.ctypes.data, _es(sens, 0), _es(sens, 1)
    else:
        sp, ss_j, ss_k = None, 0, 0
    rc = lib().wlsqm_oracle_fit_many(
        dimension, ncases, xk.ctypes.data, sx_j, sx_k, fk.ctypes.data, _es(fk, 0), _es(fk, 1),
        nk.ctypes.data, _es(nk, 0), xi.ctypes.data, sxi, fi.ctypes.data, _es(fi, 0),
        sp, ss_j, ss_k, int(bool(do_sens)), order.ctypes.data, _es(order, 0),
        knowns.ctypes.data, _es(knowns, 0), weighting_method.ctypes.data, _es(weighting_method, 0),
        int(bool(iterative)), max_iter, ntasks, C.byref(dbg) if dbg is not None else None)
    if rc < 0:
        raise ValueError("oracle fit_many failed with code %d" % rc)
    return (rc, cap) if debug_capture else rc


def max_threads():
    return lib().wlsqm_oracle_max_threads()


# ---- the checker of the ACCURATE numerics mode: oracle/variants.c with V_SYM (the oracle's arithmetic with the normal matrix
# assembled from its upper triangle and mirrored; with flags = 0 it IS the oracle, checked bit for bit by tools/attribution.py)
V_MOMENT, V_SPLIT, V_FMA, V_FASTW, V_LDLT, V_SYM = 1, 2, 4, 8, 16, 32
_VLIB = None


def _variants():
    global _VLIB
    if _VLIB is None:
        so = os.path.join(_HERE, "libwlsqm_variants.so")
        build()
        L = C.CDLL(so)
        L.wlsqm_variant_fit_many_ragged.restype = C.c_int
        L.wlsqm_variant_fit_many_ragged.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _VLIB = L
    return _VLIB


def variant_fit_many(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, flags=V_SYM, nsplit=1):
    """Uniform-order batch (2D / 3D) through oracle/variants.c: contiguous xk (n, K, dim), fk (n, K), xi (n, dim), fi (n, >= no)
    in/out, per-case nk (int32), knowns (int64), weighting_method (int32).  flags = V_SYM: the accurate mode's arithmetic."""
    n, K = fk.shape
    no = number_of_dofs(dimension, order)
    for a in (xk, fk, xi, fi):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    assert nk.dtype == np.int32 and knowns.dtype == np.int64 and weighting_method.dtype == np.int32
    assert nk.flags.c_contiguous and knowns.flags.c_contiguous and weighting_method.flags.c_contiguous
    rc = _variants().wlsqm_variant_fit_many_ragged(dimension, order, no, n, K, xk.ctypes.data, fk.ctypes.data, nk.ctypes.data,
                                                   xi.ctypes.data, fi.ctypes.data, fi.shape[1], knowns.ctypes.data,
                                                   weighting_method.ctypes.data, flags, nsplit)
    if rc != 0:
        raise ValueError("variant fit failed with code %d" % rc)
