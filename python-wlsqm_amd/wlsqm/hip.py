"""Device-resident entry points of the HIP backend (extension to the reference surface).

The reference API (wlsqm.fitter.simple / expert) takes host arrays, so every call pays PCIe for
8*nk*(dim+1) bytes per case.  These functions take arrays that already live in HBM — torch CUDA
tensors, or anything exposing ``data_ptr()``/``stride()``/``shape``/``dtype`` the same way — and
enqueue the same kernels on the caller's HIP stream with no host synchronisation.  torch is used
only as the owner of device memory and streams.
"""
import ctypes as C

from . import _binding as B

import contextlib

__all__ = ["fit_many_device", "time_fit_device", "fit_cloud_device", "time_fit_cloud_device", "device_count", "knn", "ball", "nearest",
           "last_kernel", "set_strict", "get_strict", "strict", "accurate", "strict_intermediates"]


def _mode_code(mode):
    """False / 0 -> 0 (fast), True / 1 / "strict" -> 1, 2 / "accurate" -> 2."""
    if isinstance(mode, str):
        m = mode.lower()
        if m in ("accurate", "2"):
            return 2
        if m in ("strict", "1", "true"):
            return 1
        if m in ("fast", "0", "false", ""):
            return 0
        raise ValueError("numerics mode must be 'fast', 'strict' or 'accurate', got %r" % (mode,))
    if mode is True:
        return 1
    return 2 if int(mode) == 2 else (1 if mode else 0)


def set_strict(on):
    """Numerics mode of the calling thread (wlsqm_hip_set_strict): False = the fast kernels (default, or WLSQM_HIP_STRICT in
    the environment), True / "strict" = reference-order arithmetic (csrc/fit_strict.hip: the reference's operations one for one,
    IEEE divide and sqrt, no contraction), 2 / "accurate" = the same arithmetic with the normal
    matrix assembled from its upper triangle (csrc/fit_accurate.hip: as close to the reference as the strict mode — 1e-10 on every
    column of BASELINE configs[1] / configs[4] — at a fraction of its time; cases it does not cover run the strict kernels).
    Returns the previous mode: False, True or 2."""
    prev = B.lib().wlsqm_hip_set_strict(_mode_code(on))
    return 2 if prev == 2 else bool(prev)


def get_strict():
    """False (fast), True (strict) or 2 (accurate)."""
    v = B.lib().wlsqm_hip_get_strict()
    return 2 if v == 2 else bool(v)


@contextlib.contextmanager
def strict(on=True):
    """``with wlsqm.hip.strict(): ...`` — reference-order numerics inside the block (None: leave the mode alone;
    ``strict("accurate")`` / ``strict(2)``: the accurate mode)."""
    if on is None:
        yield
        return
    prev = set_strict(on)
    try:
        yield
    finally:
        set_strict(prev)


def accurate():
    """``with wlsqm.hip.accurate(): ...`` — the accurate numerics mode inside the block."""
    return strict(2)


def device_count():
    return B.lib().wlsqm_hip_device_count()


def last_kernel():
    """Diagnostics: name of the kernel family the last fit launch of this thread dispatched to (wlsqm_hip_last_kernel)."""
    return B.lib().wlsqm_hip_last_kernel().decode()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _check(t, name, dtype_name, ndim):
    if str(t.dtype).split(".")[-1] != dtype_name:
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s' (argument %s)" % (dtype_name, t.dtype, name))
    if t.dim() != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d) (argument %s)" % (ndim, t.dim(), name))
    if not t.is_cuda:
        raise ValueError("argument %s must be a device (HIP) tensor" % name)


_NDOF = {1: (1, 2, 3, 4, 5), 2: (1, 3, 6, 10, 15), 3: (1, 4, 10, 20, 35)}


def _ndofs(dimension, order):
    if dimension not in _NDOF:
        raise ValueError("Dimension must be 1, 2 or 3")
    if not 0 <= int(order) <= 4:
        raise ValueError("order must be 0, 1, 2, 3 or 4")
    return _NDOF[dimension][int(order)]


def _rows(n, **arrays):
    """Every per-case array must cover the n cases of the launch (simple.py's _run_many makes the same checks on the host)."""
    for name, t in arrays.items():
        if t.shape[0] < n:
            raise ValueError("%s has %d rows, fewer than the %d cases of the batch" % (name, t.shape[0], n))


def _batch(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, sens, iterative, max_iter, order_dummy):
    ncases = nk.shape[0]
    no = _ndofs(dimension, order)
    _check(nk, "nk", "int32", 1); _check(knowns, "knowns", "int64", 1); _check(weighting_method, "weighting_method", "int32", 1)
    _check(fk, "fk", "float64", 2); _check(fi, "fi", "float64", 2)
    if dimension == 1:
        _check(xk, "xk", "float64", 2); _check(xi, "xi", "float64", 1)
    else:
        _check(xk, "xk", "float64", 3); _check(xi, "xi", "float64", 2)
        if xk.stride(2) != 1 or xi.stride(1) != 1:
            raise ValueError("Buffer and memoryview are not contiguous in the same dimension.")
    if fi.stride(1) != 1:
        raise ValueError("Buffer and memoryview are not contiguous in the same dimension. (argument fi)")
    # extents: the kernels write `no` doubles per fi row at the row pitch and read K neighbour slots per xk / fk row
    _rows(ncases, xk=xk, fk=fk, xi=xi, fi=fi, knowns=knowns, weighting_method=weighting_method)
    K = int(fk.shape[1])
    if xk.shape[1] < K:
        raise ValueError("xk has %d neighbour slots per case, fk has %d" % (xk.shape[1], K))
    if dimension > 1 and (xk.shape[2] < dimension or xi.shape[1] < dimension):
        raise ValueError("xk / xi must have %d coordinates on the last axis" % dimension)
    if fi.shape[1] < no:
        raise ValueError("fi has %d columns, need at least number_of_dofs(%d, %d) = %d" % (fi.shape[1], dimension, order, no))
    if sens is not None:
        _check(sens, "sens", "float64", 3)
        if sens.stride(2) != 1:
            raise ValueError("Buffer and memoryview are not contiguous in the same dimension. (argument sens)")
        if sens.shape[0] < ncases or sens.shape[1] < K or sens.shape[2] < no:
            raise ValueError("sens must be at least (ncases, %d, %d); got %s" % (K, no, tuple(sens.shape)))
    b = B.Batch()
    b.dimension, b.ncases = dimension, ncases
    b.xk, b.xk_stride_case, b.xk_stride_k = xk.data_ptr(), xk.stride(0), xk.stride(1)
    b.fk, b.fk_stride_case, b.fk_stride_k = fk.data_ptr(), fk.stride(0), fk.stride(1)
    b.nk, b.nk_stride = nk.data_ptr(), nk.stride(0)
    b.xi, b.xi_stride_case = xi.data_ptr(), xi.stride(0)
    b.fi, b.fi_stride_case = fi.data_ptr(), fi.stride(0)
    if sens is not None:
        b.do_sens = 1
        b.sens, b.sens_stride_case, b.sens_stride_k = sens.data_ptr(), sens.stride(0), sens.stride(1)
    # the per-case order array is not read on this path (order_uniform is); point it somewhere valid
    b.order, b.order_stride = order_dummy.data_ptr(), 0
    b.knowns, b.knowns_stride = knowns.data_ptr(), knowns.stride(0)
    b.weighting_method, b.wm_stride = weighting_method.data_ptr(), weighting_method.stride(0)
    b.iterative, b.max_iter, b.max_nk = (1 if iterative else 0), int(max_iter), int(fk.shape[1])
    return b


def _strict_ctx(flag):
    return strict(flag)


def _stream_and_device(t, stream):
    import torch
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    return C.c_void_p(int(stream) if stream else 0), dev


class row_hint:
    """with wlsqm.hip.row_hint("ragged"): ... — what the calling thread knows about the neighbour counts of the dense device-resident batches
    it hands over (wlsqm_hip_set_row_hint): "full" (default: every case fills its row), "ragged" (e.g. ball-query rows: the staged kernels'
    waves move only the chunks their own cases need) or None / "unknown" (the kernels find out: one idle launch for full rows).  A hint: the
    results are the same bits either way."""
    _CODE = {"full": 1, True: 2, "ragged": 2, None: 0, "unknown": 0, False: 1}

    def __init__(self, what="full", sorted=True):
        # sorted: the neighbours of every row come sorted by distance (a k-nearest-neighbour search's rows: the default) or not (a ball
        # query's: sorted=False) — wlsqm_hip_set_order_hint: picks the form of the staged kernels of the small dense systems
        self.code = self._CODE[what]
        self.sorted = 1 if sorted else 0

    def __enter__(self):
        L = B.lib()
        self.prev = L.wlsqm_hip_set_row_hint(self.code) if hasattr(L, "wlsqm_hip_set_row_hint") else 1
        self.prev_sorted = L.wlsqm_hip_set_order_hint(self.sorted) if hasattr(L, "wlsqm_hip_set_order_hint") else 1
        return self

    def __exit__(self, *exc):
        L = B.lib()
        if hasattr(L, "wlsqm_hip_set_row_hint"):
            L.wlsqm_hip_set_row_hint(self.prev)
        if hasattr(L, "wlsqm_hip_set_order_hint"):
            L.wlsqm_hip_set_order_hint(self.prev_sorted)
        return False


def fit_many_device(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, sens=None, iterative=False,
                    max_iter=10, case_index=None, stream=None, want_iterations=False, strict=None, max_order=4):
    """fit_{1,2,3}D[_iterative]_many on device-resident tensors.  `order`: an int (all cases of that polynomial order: the fast
    path) or an int32 device tensor of per-case orders (fi / sens must then be wide enough for `max_order`, default 4: the
    reference's "(ncases, >= max no)" rule with the maximum taken from the caller instead of a host scan of the tensor).
    strict=True / False selects reference-order / fast numerics for this call (None: the thread's mode, see set_strict).

    xk (n, K, dim) [1D: (n, K)], fk (n, K), nk (n,) int32, xi (n, dim) [1D: (n,)], fi (n, >=no) in/out,
    knowns (n,) int64, weighting_method (n,) int32, sens (n, K, >=no) or None.  `case_index` (int64 device
    tensor) restricts the launch to those cases (used to bucket heterogeneous orders).  Asynchronous on
    `stream` (default: torch's current stream) unless want_iterations=True.  The caller guarantees that
    fk/xk do not alias fi (outputs are written in place)."""
    if hasattr(order, "data_ptr"):
        # per-case orders (int32 device tensor, as the reference's `order` array): bucketed on the device, no host synchronisation
        _check(order, "order", "int32", 1)
        if case_index is not None:
            raise ValueError("case_index and a per-case order tensor exclude each other")
        _rows(nk.shape[0], order=order)
        # extents are checked for `max_order` (default 4): the caller's promise that no case asks for more
        b = _batch(dimension, int(max_order), xk, fk, nk, xi, fi, knowns, weighting_method, sens, iterative, max_iter, nk)
        s, dev = _stream_and_device(fi, stream)
        its = C.c_int32(0)
        with _strict_ctx(strict):
            B.check(B.lib().wlsqm_hip_fit_many_device_orders(C.byref(b), dev, s, C.c_void_p(order.data_ptr()), order.stride(0),
                                                             int(max_order), C.byref(its) if want_iterations else None))
        return int(its.value)
    b = _batch(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, sens, iterative, max_iter, nk)
    s, dev = _stream_and_device(fi, stream)
    its = C.c_int32(0)
    nsel = 0
    ci = None
    if case_index is not None:
        _check(case_index, "case_index", "int64", 1)
        ci, nsel = C.c_void_p(case_index.data_ptr()), int(case_index.shape[0])
    with _strict_ctx(strict):
        B.check(B.lib().wlsqm_hip_fit_many_device(C.byref(b), dev, s, int(order), ci, nsel,
                                                  C.byref(its) if want_iterations else None))
    return int(its.value)


def strict_intermediates(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, stream=None):
    """Test hook (wlsqm_hip_strict_intermediates_device): the reference-order fit of a uniform-order batch, which also returns
    the reference's intermediates as device tensors: dict(w (n, K), A (n, no*no), LU (n, no*no), row_scale (n, no),
    col_scale (n, no), ipiv (n, no) int32).  A / LU hold the nr x nr Fortran-order block of each case at the front of its row."""
    import torch
    b = _batch(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, None, False, 0, nk)
    s, dev = _stream_and_device(fi, stream)
    n, K, no = int(nk.shape[0]), int(fk.shape[1]), _ndofs(dimension, order)
    z = lambda *shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=fi.device)
    out = dict(w=z(n, K), A=z(n, no * no), LU=z(n, no * no), row_scale=z(n, no), col_scale=z(n, no), ipiv=z(n, no, dt=torch.int32))
    B.check(B.lib().wlsqm_hip_strict_intermediates_device(C.byref(b), dev, s, int(order), _ptr(out["w"]), K, _ptr(out["A"]),
                                                          _ptr(out["LU"]), no * no, _ptr(out["row_scale"]), _ptr(out["col_scale"]),
                                                          _ptr(out["ipiv"]), no))
    return out


def time_fit_device(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, reps=10, stream=None):
    """Mean duration in milliseconds of one fit launch (HIP events on `stream`, `reps` back-to-back launches)."""
    b = _batch(dimension, order, xk, fk, nk, xi, fi, knowns, weighting_method, None, False, 0, nk)
    s, dev = _stream_and_device(fi, stream)
    ms = C.c_float(0.0)
    B.check(B.lib().wlsqm_hip_time_fit_device(C.byref(b), dev, s, int(order), int(reps), C.byref(ms)))
    return float(ms.value)


def _cloud_args(dimension, order, S, F, hoods, fi, nk, knowns, weighting_method, point_index):
    _check(F, "F", "float64", 1); _check(hoods, "hoods", "int32", 2); _check(fi, "fi", "float64", 2)
    _check(S, "S", "float64", 1 if dimension == 1 else 2)
    _check(nk, "nk", "int32", 1); _check(knowns, "knowns", "int64", 1); _check(weighting_method, "weighting_method", "int32", 1)
    if not S.is_contiguous() or not F.is_contiguous() or hoods.stride(1) != 1 or fi.stride(1) != 1:
        raise ValueError("S, F must be contiguous; hoods and fi must have a contiguous last axis")
    for t in (nk, knowns, weighting_method):
        if t.stride(0) != 1:
            raise ValueError("nk, knowns, weighting_method must have unit stride")
    if point_index is not None:
        _check(point_index, "point_index", "int32", 1)
    ncases, K = int(hoods.shape[0]), int(hoods.shape[1])
    no = _ndofs(dimension, order)
    _rows(ncases, fi=fi, nk=nk, knowns=knowns, weighting_method=weighting_method)
    if point_index is not None:
        _rows(ncases, point_index=point_index)
    elif S.shape[0] < ncases:
        raise ValueError("S has %d points but there are %d cases (xi of case j is S[j] without point_index)" % (S.shape[0], ncases))
    if dimension > 1 and S.shape[1] != dimension:
        raise ValueError("S must be (npoints, %d)" % dimension)
    if F.shape[0] < S.shape[0]:
        raise ValueError("F has fewer entries than S has points")
    if fi.shape[1] < no:
        raise ValueError("fi has %d columns, need at least number_of_dofs(%d, %d) = %d" % (fi.shape[1], dimension, order, no))
    return [int(dimension), int(order), ncases, K, _ptr(S), _ptr(F), _ptr(hoods), int(hoods.stride(0)), _ptr(point_index),
            _ptr(nk), _ptr(knowns), _ptr(weighting_method), _ptr(fi), int(fi.stride(0))]


def fit_cloud_device(dimension, order, S, F, hoods, fi, nk, knowns, weighting_method, point_index=None, sens=None,
                     iterative=False, max_iter=10, stream=None, want_iterations=False, strict=None):
    """Index-based fit (extension): the kernels gather xk = S[hoods], fk = F[hoods] themselves.

    S (npoints, dim) [1D: (npoints,)] and F (npoints,) are the device-resident point tables, hoods (ncases, K)
    int32 the neighbour lists, xi of case j is S[point_index[j]] (default: S[j]); nk/knowns/weighting_method
    per case; fi (ncases, >= no) in/out.  Only the slots k < nk[j] of a hoods row are dereferenced (the padding of a ragged
    row may hold anything, e.g. -1 or npoints as scipy pads); those must be valid point numbers.  Asynchronous on `stream`
    unless want_iterations=True (returns the maximum refinement count; otherwise 0).  4 nk bytes of indices per fit instead of 8 nk (dim+1) bytes of gathered
    coordinates; order the points along a space-filling curve (synth.morton_order) so that the gathers hit L2."""
    a = _cloud_args(dimension, order, S, F, hoods, fi, nk, knowns, weighting_method, point_index)
    s, dev = _stream_and_device(fi, stream)
    ss = (int(sens.stride(0)), int(sens.stride(1))) if sens is not None else (0, 0)
    if sens is not None:
        _check(sens, "sens", "float64", 3)
        if sens.stride(2) != 1 or sens.shape[0] < hoods.shape[0] or sens.shape[1] < hoods.shape[1] \
                or sens.shape[2] < _ndofs(dimension, order):
            raise ValueError("sens must be at least (ncases, max_nk, no) with a contiguous last axis; got %s" % (tuple(sens.shape),))
    its = C.c_int32(0)
    with _strict_ctx(strict):
        B.check(B.lib().wlsqm_hip_fit_cloud_device(*a, _ptr(sens), ss[0], ss[1], 1 if sens is not None else 0,
                                                   1 if iterative else 0, int(max_iter), dev, s,
                                                   C.byref(its) if want_iterations else None))
    return int(its.value)


def time_fit_cloud_device(dimension, order, S, F, hoods, fi, nk, knowns, weighting_method, point_index=None, reps=10,
                          stream=None):
    """Mean duration in milliseconds of one index-based fit launch (HIP events on `stream`)."""
    a = _cloud_args(dimension, order, S, F, hoods, fi, nk, knowns, weighting_method, point_index)
    s, dev = _stream_and_device(fi, stream)
    ms = C.c_float(0.0)
    B.check(B.lib().wlsqm_hip_time_fit_cloud_device(*a, dev, s, int(reps), C.byref(ms)))
    return float(ms.value)


def knn(S, k, stream=None, nquery=None):
    """The k nearest OTHER points of every point of the device-resident cloud S (npoints, dim) [1D: (npoints,)], as an
    int32 device tensor (npoints, k), ascending by (distance, index) — what the reference's examples get on the host from
    ``cKDTree(S).query(S, 1 + k)[1][:, 1:]`` (examples/expertsolver_example.py:48-66).  Exact uniform-grid search on the
    GPU; the result is the ``hoods`` argument of fit_cloud_device / ShardedCloudSolver, or of ``S[hoods]`` for the dense
    API.  1 <= k <= min(npoints - 1, 213).  Synchronises the stream.
    With `nquery`, only the first nquery points ask and the rest are candidates only: returns (nquery, k)."""
    import torch
    if S.dim() == 1:
        dim = 1
    elif S.dim() == 2 and 1 <= S.shape[1] <= 3:
        dim = int(S.shape[1])
    else:
        raise ValueError("S must be (npoints,) or (npoints, dim) with dim 1..3")
    _check(S, "S", "float64", S.dim())
    if not S.is_contiguous():
        raise ValueError("S must be contiguous")
    n = int(S.shape[0])
    nq = n if nquery is None else int(nquery)
    if not 0 < nq <= n:
        raise ValueError("nquery must be in 1 .. npoints")
    hoods = torch.empty((nq, int(k)), dtype=torch.int32, device=S.device)
    s, dev = _stream_and_device(S, stream)
    if nquery is None:
        B.check(B.lib().wlsqm_hip_knn_device(dim, n, _ptr(S), int(k), _ptr(hoods), dev, s))
    else:
        B.check(B.lib().wlsqm_hip_knn_subset_device(dim, n, _ptr(S), int(k), nq, _ptr(hoods), dev, s))
    return hoods


def ball(S, radius, max_nk, stream=None):
    """All OTHER points within `radius` of every point of the device-resident cloud S, nearest first, at most max_nk of
    them: returns (hoods, nk) = int32 device tensors (npoints, max_nk) and (npoints,).  The radius form of knn(): what
    the reference's examples/wlsqm_example.py:103-133 builds with ``cKDTree.query_ball_point`` (unused slots of a row
    hold the point's own index, so the rows stay valid for fit_cloud_device with the returned nk)."""
    import torch
    dim = 1 if S.dim() == 1 else int(S.shape[1])
    if S.dim() not in (1, 2) or not 1 <= dim <= 3:
        raise ValueError("S must be (npoints,) or (npoints, dim) with dim 1..3")
    _check(S, "S", "float64", S.dim())
    if not S.is_contiguous():
        raise ValueError("S must be contiguous")
    n = int(S.shape[0])
    hoods = torch.empty((n, int(max_nk)), dtype=torch.int32, device=S.device)
    nk = torch.empty((n,), dtype=torch.int32, device=S.device)
    s, dev = _stream_and_device(S, stream)
    B.check(B.lib().wlsqm_hip_ball_device(dim, n, _ptr(S), float(radius), int(max_nk), _ptr(hoods), _ptr(nk), dev, s))
    return hoods, nk


def nearest(S, X, stream=None):
    """Index (int64 device tensor, one per row of X) of the point of the device-resident cloud S nearest to each query
    point X[j] (queries need not belong to the cloud; ties go to the smaller index): ``cKDTree(S).query(X)[1]`` on the GPU."""
    import torch
    dim = 1 if S.dim() == 1 else int(S.shape[1])
    _check(S, "S", "float64", S.dim()); _check(X, "X", "float64", X.dim())
    if not S.is_contiguous() or not X.is_contiguous() or (1 if X.dim() == 1 else int(X.shape[1])) != dim:
        raise ValueError("S and X must be contiguous and have the same number of coordinates")
    out = torch.empty((int(X.shape[0]),), dtype=torch.int64, device=S.device)
    s, dev = _stream_and_device(S, stream)
    B.check(B.lib().wlsqm_hip_nearest_device(dim, int(S.shape[0]), _ptr(S), int(X.shape[0]), _ptr(X), dim, _ptr(out), dev, s))
    return out
