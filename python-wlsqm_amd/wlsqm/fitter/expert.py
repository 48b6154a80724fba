"""Advanced API ("expert mode" in the LAPACK sense): separate prepare and solve stages.

Drop-in mirror of the reference's wlsqm/fitter/expert.pyx (ExpertSolver, expert.pyx:66-655,
and number_of_dofs, expert.pyx:57-63), backed by the MI355X HIP kernels.  The prepared
geometry lives in HBM for the lifetime of the solver; ``solve()`` streams ``fk`` in and the
coefficient array ``fi`` in/out.
"""
import ctypes as C

import numpy as np

from . import defs
from .. import _binding as B

__all__ = ["number_of_dofs", "ExpertSolver"]


def number_of_dofs(dimension, order):
    """Number of DOFs of the full polynomial for dimension (1,2,3) and order (0..4) (expert.pyx:57-63)."""
    return B.lib().wlsqm_hip_number_of_dofs(int(dimension), int(order))


def _to_numpy(a):
    """host copy of a numpy array or of a (device) torch tensor"""
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


class ExpertSolver:
    """Prepare once (geometry), solve many times (data).  See the reference's expert.pyx:66-90.

    s = ExpertSolver(...); s.prepare(xi, xk); s.solve(fk, fi[, sens])
    """

    def __init__(self, dimension, nk, order, knowns, weighting_method, algorithm=defs.ALGO_BASIC, do_sens=False,
                 max_iter=10, ntasks=1, debug=False, host=None):
        self._handle = None
        self._tree = None
        self._tree_points = None
        nk = B.view(nk, np.int32, 1, "nk")
        order = B.view(order, np.int32, 1, "order")
        knowns = B.view(knowns, np.int64, 1, "knowns")
        weighting_method = B.view(weighting_method, np.int32, 1, "weighting_method")
        ncases = nk.shape[0]
        if order.shape[0] != ncases or knowns.shape[0] != ncases or weighting_method.shape[0] != ncases:   # expert.pyx:131-132
            raise ValueError("nk, order, knowns and weighting method must have the same length; currently, "
                             "len(nk)=%d, len(order)=%d, len(knowns)=%d, len(weighting_method)=%d"
                             % (nk.shape[0], order.shape[0], knowns.shape[0], weighting_method.shape[0]))
        if dimension not in (1, 2, 3):                                                                    # :134-135
            raise ValueError("Dimension must be 1, 2 or 3, got %d" % dimension)
        for name, v in (("algorithm", algorithm), ("do_sens", do_sens), ("max_iter", max_iter),
                        ("ntasks", ntasks), ("debug", debug)):                                             # :139-148
            if v is None:
                raise ValueError("%s cannot be None" % name)
        if algorithm not in (defs.ALGO_BASIC, defs.ALGO_ITERATIVE):                                      # :151-156
            raise ValueError("Unknown algorithm specifier %d; see wlsqm.fitter.defs for valid specifiers ALGO_*" % algorithm)
        if ntasks < 1:                                                                                    # :158-159
            raise ValueError("ntasks must be >= 1, got %d" % ntasks)
        if host is not None:                                                                              # :163-189
            if not host.ready:
                raise RuntimeError("In guest mode, host must be in the ready state (host.prepare() must have been "
                                   "called before creating another ExpertSolver instance in guest mode).")
            if host.ncases != ncases:
                raise RuntimeError("In guest mode, number of cases (number of elements in nk) must match; got %d, host has %d" % (ncases, host.ncases))
            if host.dimension != dimension:
                raise ValueError("In guest mode, dimension must match; got %d, host has %d" % (dimension, host.dimension))
            if bool(host.debug) != bool(debug):
                raise ValueError("In guest mode, debug flag must match; got %s, host has %s" % (bool(debug), bool(host.debug)))
            for name, mine, theirs in (("nk", nk, host.nk), ("order", order, host.order), ("knowns", knowns, host.knowns),
                                       ("weighting_method", weighting_method, host.weighting_method)):
                if (np.asanyarray(theirs) != np.asanyarray(mine)).any():
                    raise ValueError("In guest mode, '%s' must match element-by-element." % name)
        self.host = host
        self.ready = False
        self.dimension, self.algorithm, self.max_iter = int(dimension), int(algorithm), int(max_iter)
        self.ncases, self.do_sens, self.ntasks, self.debug = ncases, bool(do_sens), int(ntasks), bool(debug)
        self.xk = self.xi = self.tree = None
        self.nk, self.order, self.knowns, self.weighting_method = nk, order, knowns, weighting_method
        self._max_nk = int(nk.max()) if ncases else 0
        self._max_no = number_of_dofs(self.dimension, int(np.max(order))) if ncases else 0
        self._fi_device = None
        h = C.c_void_p()
        if host is not None:
            # guest mode: share the host's device-resident geometry and metadata, own only the field buffers
            if not getattr(host, "_handle", None):
                raise RuntimeError("In guest mode, the host ExpertSolver has been closed")
            B.check(B.lib().wlsqm_hip_expert_create_guest(C.byref(h), host._handle, self.algorithm,
                                                          int(self.do_sens), self.max_iter))
        else:
            B.check(B.lib().wlsqm_hip_expert_create(
                C.byref(h), B.default_device(), self.dimension, ncases,
                np.ascontiguousarray(nk).ctypes.data, np.ascontiguousarray(order).ctypes.data,
                np.ascontiguousarray(knowns).ctypes.data, np.ascontiguousarray(weighting_method).ctypes.data,
                self.algorithm, int(self.do_sens), self.max_iter))
        self._handle = h
        if host is not None:
            self._tree, self._tree_points = host._tree, host._tree_points        # expert.pyx:263-264

    def close(self):
        """Release the device-side state now (also done by __del__; expert.pyx:267-286)."""
        if getattr(self, "_handle", None):
            B.lib().wlsqm_hip_expert_destroy(self._handle)
            self._handle = None
            self.ready = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- persistence (extension: copying / pickling are on the reference's TODO list).  The device-side state is a
    # function of the constructor arguments and of the geometry handed to prepare(), both of which the object keeps, so a
    # copy or an unpickled solver is rebuilt from them; the fitted coefficients of the last solve() are not carried over.
    def __getstate__(self):
        return dict(dimension=self.dimension, nk=np.asarray(self.nk), order=np.asarray(self.order),
                    knowns=np.asarray(self.knowns), weighting_method=np.asarray(self.weighting_method),
                    algorithm=self.algorithm, do_sens=self.do_sens, max_iter=self.max_iter, ntasks=self.ntasks,
                    debug=self.debug, host=self.host, ready=self.ready,
                    xi=None if self.xi is None else _to_numpy(self.xi), xk=None if self.xk is None else _to_numpy(self.xk))

    def __setstate__(self, st):
        ready, xi, xk = st.pop("ready"), st.pop("xi"), st.pop("xk")
        self.__init__(**st)
        if ready:
            self.prepare(xi=xi, xk=xk)

    def __copy__(self):
        new = ExpertSolver.__new__(ExpertSolver)
        new.__setstate__(self.__getstate__())
        return new

    def __deepcopy__(self, memo):
        import copy
        st = self.__getstate__()
        st["host"] = copy.deepcopy(st["host"], memo)
        new = ExpertSolver.__new__(ExpertSolver)
        new.__setstate__(st)
        return new

    def memory_used(self):
        """(bytes in use, bytes reserved) of the device-side state (expert.pyx:289-306)."""
        used, total = C.c_int64(0), C.c_int64(0)
        B.check(B.lib().wlsqm_hip_expert_memory_used(self._handle, C.byref(used), C.byref(total)))
        return (int(used.value), int(total.value))

    def prepare(self, xi, xk):
        """Upload the geometry (expert.pyx:309-426).  In guest mode the host's geometry is used (:350-352): nothing is
        uploaded, the arguments are ignored and the host's device-resident copy is shared."""
        self.ready = False
        if self.host is not None:
            B.check(B.lib().wlsqm_hip_expert_prepare(self._handle, None, 0, None, 0, 0, 0))
            self.xk, self.xi = self.host.xk, self.host.xi
            self.ready = True
            return
        if self.dimension == 1:
            xiv = B.view(xi, np.float64, 1, "xi")
            xkv = B.view(xk, np.float64, 2, "xk")
        else:
            xiv = B.view(xi, np.float64, 2, "xi", contiguous_last=True)
            xkv = B.view(xk, np.float64, 3, "xk", contiguous_last=True)
            if xiv.shape[1] < self.dimension or xkv.shape[2] < self.dimension:
                raise ValueError("xi/xk must have %d coordinates on the last axis" % self.dimension)
        if xiv.shape[0] < self.ncases or xkv.shape[0] < self.ncases:
            raise ValueError("xi/xk have fewer rows than ncases")
        if xkv.shape[1] < self._max_nk:
            raise ValueError("max(nk) = %d exceeds the neighbour axis of xk (%d)" % (self._max_nk, xkv.shape[1]))
        B.check(B.lib().wlsqm_hip_expert_prepare(self._handle, xiv.ctypes.data, B.es(xiv, 0), xkv.ctypes.data,
                                                 B.es(xkv, 0), B.es(xkv, 1), xkv.shape[1]))
        self.xk, self.xi = xk, xi                                                                      # :387-388
        if self.host is None:
            self.tree = None
        self.ready = True

    def prepare_device(self, xi, xk, stream=None):
        """prepare() from device-resident torch tensors (extension): xi (ncases, dim) [1D: (ncases,)], xk (ncases, >= max_nk,
        dim) [1D: (ncases, >= max_nk)], float64, contiguous last axis.  The geometry is copied device-to-device into the
        solver's own block (no PCIe traffic); together with wlsqm.hip.knn / ball and solve_device the whole workflow stays
        on the GPU."""
        self.ready = False
        if self.host is not None:
            return self.prepare(None, None)
        import torch
        want = (1, 2) if self.dimension == 1 else (2, 3)
        if xi.dim() != want[0] or xk.dim() != want[1] or xi.dtype != torch.float64 or xk.dtype != torch.float64 \
                or not xi.is_cuda or not xk.is_cuda:
            raise ValueError("xi / xk must be float64 device tensors of %d / %d dimensions" % want)
        if self.dimension > 1 and (xi.stride(1) != 1 or xk.stride(2) != 1 or xk.shape[2] != self.dimension
                                   or xi.shape[1] != self.dimension or xk.stride(1) != self.dimension):
            raise ValueError("xi (ncases, dim) and xk (ncases, K, dim) must have contiguous rows of exactly `dimension` coordinates")
        if self.dimension == 1 and xk.stride(1) != 1:
            raise ValueError("xk (ncases, K) must have a contiguous neighbour axis")
        if xi.shape[0] < self.ncases or xk.shape[0] < self.ncases or xk.shape[1] < self._max_nk:
            raise ValueError("xi / xk are too small")
        if stream is None:
            stream = torch.cuda.current_stream(xk.device).cuda_stream
        B.check(B.lib().wlsqm_hip_expert_prepare_device(self._handle, C.c_void_p(int(stream) if stream else 0),
                                                        C.c_void_p(xi.data_ptr()), xi.stride(0), C.c_void_p(xk.data_ptr()),
                                                        xk.stride(0), self.dimension))
        self.xk, self.xi = xk, xi
        self.tree = None
        self.ready = True

    def conds(self):
        """2-norm condition numbers of the (Ruiz-scaled) problem matrices, shape (ncases,) (expert.pyx:429-464).
        Only available in debug mode, like the reference (which fills them during prepare(), impl.pyx:662-682);
        here they are computed on the device from the resident geometry when asked for."""
        if not self.ready:
            raise RuntimeError("Solver is not in the ready state; prepare() must be called before conds()")
        if not self.debug:
            raise RuntimeError("Not in debug mode; condition number data has not been computed")
        out = np.empty((self.ncases,), dtype=np.float64)
        B.check(B.lib().wlsqm_hip_expert_conds(self._handle, out.ctypes.data))
        return out

    # `tree` (expert.pyx:247, 681: a scipy cKDTree of the origins xi) is built on first use: mode='nearest' finds the nearest
    # origin on the GPU and never needs it; mode='continuous' (ball queries against the origins) does.
    @property
    def tree(self):
        if self._tree is None and self._tree_points is not None:
            import scipy.spatial
            self._tree = scipy.spatial.cKDTree(data=self._tree_points)
        return self._tree

    @tree.setter
    def tree(self, value):
        self._tree = value
        if value is None:
            self._tree_points = None

    def prep_interpolate(self):
        """Prepare interpolate() (expert.pyx:658-681: index the origins xi for the nearest-model search).  The k-d tree of
        the reference is only built when a caller asks for `self.tree`; both interpolation modes search on the GPU."""
        if not self.ready:
            raise RuntimeError("Solver is not in the ready state; prepare() must be called before prep_interpolate()")
        if self.host is not None and self.host._tree_points is not None:
            self._tree, self._tree_points = self.host._tree, self.host._tree_points
        else:
            xi = _to_numpy(self.xi)
            self._tree, self._tree_points = None, (xi if self.dimension >= 2 else np.atleast_2d(xi).T)

    def interpolate(self, x, mode='nearest', r=None, diff=0, I=None):
        """Interpolate the global (patched) model or its derivative `diff` to the points x (expert.pyx:687-781).

        mode='nearest': use the local model whose origin is nearest (or the models named by I); returns
        (out, I_out).  mode='continuous': weighted average of all local models within radius r; returns (out, None)."""
        if mode not in ('nearest', 'continuous'):
            raise ValueError("mode must be one of 'nearest', 'continuous'; got '%s'" % (mode,))
        if mode == 'continuous' and r is None:
            raise ValueError("r must be specified in mode='continuous'")
        if diff is None:
            raise ValueError("diff cannot be None")
        if self._tree is None and self._tree_points is None:
            raise RuntimeError("Points xi have not been indexed; prep_interpolate() must be called before interpolate()")
        if I is not None and len(I) != len(x):
            raise ValueError("When 'I' is specified, 'I' must have the same length as x; got len(I) = %d, len(x) = %d." % (len(I), len(x)))
        if self.dimension == 1:
            xv = B.view(x, np.float64, 1, "x")
        else:
            xv = B.view(x, np.float64, 2, "x", contiguous_last=True)
        nx = xv.shape[0]
        out = np.empty((nx,), dtype=np.float64)
        lib = B.lib()
        if mode == 'nearest':
            if I is None:                                                        # nearest origin found on the device
                I_out = np.empty((nx,), dtype=np.int64)
                B.check(lib.wlsqm_hip_expert_interpolate_nearest(self._handle, xv.ctypes.data, B.es(xv, 0), nx, int(diff),
                                                                 out.ctypes.data, I_out.ctypes.data))
                return out, I_out
            I_out = I
            Iv = np.ascontiguousarray(np.asarray(I_out, dtype=np.int64))
            if (Iv == self.ncases).any():                                        # expert.pyx:861-864
                out[:] = np.nan
                return out, np.asanyarray(I_out)
            B.check(lib.wlsqm_hip_expert_interpolate(self._handle, xv.ctypes.data, B.es(xv, 0), nx, Iv.ctypes.data,
                                                     None, None, 0.0, int(diff), out.ctypes.data))
            return out, np.asanyarray(I_out)
        # the models within r are found and averaged on the device (expert.pyx:901-903 builds lists with query_ball_tree)
        B.check(lib.wlsqm_hip_expert_interpolate_continuous(self._handle, xv.ctypes.data, B.es(xv, 0), nx, float(r), int(diff),
                                                            out.ctypes.data))
        return out, np.asanyarray(None)

    def solve_device(self, fk, fi, stream=None):
        """Device-resident solve (extension): fk (ncases, max_nk) and fi (ncases, >= no) are torch CUDA tensors
        (float64, contiguous last axis); the fit is enqueued on `stream` (default: torch's current stream) with no
        host synchronisation and no PCIe traffic.  ALGO_BASIC, no sensitivities.  This is the time-stepping fast
        path: fi can feed the next step's fk without leaving HBM."""
        if not self.ready:
            raise RuntimeError("Solver is not in the ready state; prepare() must be called before solve()")
        import torch
        for t, name in ((fk, "fk"), (fi, "fi")):
            if t.dtype != torch.float64 or t.dim() != 2 or not t.is_cuda or t.stride(1) != 1:
                raise ValueError("%s must be a 2-D float64 device tensor with a contiguous last axis" % name)
        if fk.shape[0] < self.ncases or fi.shape[0] < self.ncases or fk.shape[1] < self._max_nk:
            raise ValueError("fk/fi are too small")
        if fi.shape[1] < self._max_no:
            raise ValueError("fi has %d columns, need at least %d" % (fi.shape[1], self._max_no))
        if stream is None:
            stream = torch.cuda.current_stream(fi.device).cuda_stream
        self._fi_device = fi            # interpolate() evaluates the latest solve: keep its coefficients alive
        B.check(B.lib().wlsqm_hip_expert_solve_device(self._handle, C.c_void_p(int(stream) if stream else 0),
                                                      C.c_void_p(fk.data_ptr()), fk.stride(0),
                                                      C.c_void_p(fi.data_ptr()), fi.stride(0)))
        return 0

    def prepare_operator(self, stream=None):
        """Build the stored solution operator of the prepared geometry now (extension): the stacked solves of 64 fields or more
        (and every stacked solve of a shape with more than 6 unknowns or 32 neighbour slots) apply it as one batched GEMM on the
        matrix cores; without this call the first such solve builds it and synchronises.  Returns True when the operator exists
        afterwards (False: the shape has none, or it does not fit the free device memory — the other kernels are used)."""
        if not self.ready:
            raise RuntimeError("Solver is not in the ready state; prepare() must be called before solve()")
        built = C.c_int(0)
        if stream is None:
            try:
                import torch
                stream = torch.cuda.current_stream().cuda_stream
            except Exception:
                stream = 0
        B.check(B.lib().wlsqm_hip_expert_prepare_operator(self._handle, C.c_void_p(int(stream) if stream else 0), C.byref(built)))
        return bool(built.value)

    def solve_many_device(self, fk, fi, stream=None):
        """Many fields on the prepared geometry, device-resident (extension; no reference counterpart).

        fk (nrhs, ncases, max_nk) and fi (nrhs, ncases, >= no) are torch CUDA tensors (float64, contiguous last axis);
        field r is fitted from fk[r] into fi[r] exactly as solve_device(fk[r], fi[r]) would (ALGO_BASIC, no
        sensitivities; knowns are read from fi[r]), but in ONE launch: per case and field only fk[r] is read and fi[r] written.
        Stacks of 64 fields or more, and every stack on a shape with more than 6 unknowns or 32 neighbour slots, apply the stored
        solution operator (prepare_operator(); built by the first such call otherwise) as a batched GEMM on the matrix cores; short
        stacks on small shapes share the geometry work inside the launch instead."""
        if not self.ready:
            raise RuntimeError("Solver is not in the ready state; prepare() must be called before solve()")
        import torch
        for t, name in ((fk, "fk"), (fi, "fi")):
            if t.dtype != torch.float64 or t.dim() != 3 or not t.is_cuda or t.stride(2) != 1:
                raise ValueError("%s must be a 3-D float64 device tensor with a contiguous last axis" % name)
        if fk.shape[0] != fi.shape[0] or fk.shape[0] < 1:
            raise ValueError("fk and fi must hold the same number (>= 1) of right-hand sides")
        if fk.shape[1] < self.ncases or fi.shape[1] < self.ncases or fk.shape[2] < self._max_nk:
            raise ValueError("fk/fi are too small")
        if fi.shape[2] < self._max_no:
            raise ValueError("fi has %d columns, need at least %d" % (fi.shape[2], self._max_no))
        self._fi_device = fi            # interpolate() evaluates the latest solve (here: the last field)
        if stream is None:
            stream = torch.cuda.current_stream(fi.device).cuda_stream
        B.check(B.lib().wlsqm_hip_expert_solve_many_device(
            self._handle, C.c_void_p(int(stream) if stream else 0), fk.shape[0],
            C.c_void_p(fk.data_ptr()), fk.stride(0), fk.stride(1), C.c_void_p(fi.data_ptr()), fi.stride(0), fi.stride(1)))
        return 0

    def solve_many(self, fk, fi):
        """Many fields on the prepared geometry, numpy in/out (extension): fk (nrhs, ncases, >= max_nk),
        fi (nrhs, ncases, >= no) in/out; same result as nrhs calls of solve() with ALGO_BASIC.  Returns 0."""
        if not self.ready:
            raise RuntimeError("Solver is not in the ready state; prepare() must be called before solve()")
        fkv = B.view(fk, np.float64, 3, "fk")
        fiv = B.view(fi, np.float64, 3, "fi", contiguous_last=True, writable=True)
        if fkv.shape[0] != fiv.shape[0] or fkv.shape[0] < 1:
            raise ValueError("fk and fi must hold the same number (>= 1) of right-hand sides")
        if fkv.shape[1] < self.ncases or fiv.shape[1] < self.ncases:
            raise ValueError("fk/fi have fewer rows than ncases")
        if fkv.shape[2] < self._max_nk:
            raise ValueError("max(nk) = %d exceeds the neighbour axis of fk" % self._max_nk)
        B.check(B.lib().wlsqm_hip_expert_solve_many(
            self._handle, fkv.shape[0], fkv.ctypes.data, B.es(fkv, 0), B.es(fkv, 1), B.es(fkv, 2),
            fiv.ctypes.data, B.es(fiv, 0), B.es(fiv, 1)))
        return 0

    def solve(self, fk, fi, sens=None):
        """Fit all cases to the data fk on the prepared geometry (expert.pyx:467-655).  Returns the maximum
        number of refinement iterations taken (0 for ALGO_BASIC)."""
        if not self.ready:
            raise RuntimeError("Solver is not in the ready state; prepare() must be called before solve()")
        fkv = B.view(fk, np.float64, 2, "fk")
        fiv = B.view(fi, np.float64, 2, "fi", contiguous_last=True, writable=True)
        if fkv.shape[0] < self.ncases or fiv.shape[0] < self.ncases:
            raise ValueError("fk/fi have fewer rows than ncases")
        if fkv.shape[1] < self._max_nk:
            raise ValueError("max(nk) = %d exceeds the neighbour axis of fk" % self._max_nk)
        max_no = number_of_dofs(self.dimension, int(np.max(self.order)))
        if fiv.shape[1] < max_no:
            raise ValueError("fi has %d columns, need at least %d" % (fiv.shape[1], max_no))
        sp, ssj, ssk = None, 0, 0
        if self.do_sens:
            if sens is None:
                raise ValueError("do_sens is set but sens is None")
            sv = B.view(sens, np.float64, 3, "sens", contiguous_last=True, writable=True)
            if sv.shape[0] < self.ncases or sv.shape[1] < self._max_nk or sv.shape[2] < max_no:
                raise ValueError("sens is too small")
            sp, ssj, ssk = sv.ctypes.data, B.es(sv, 0), B.es(sv, 1)
        its = C.c_int32(0)
        B.check(B.lib().wlsqm_hip_expert_solve(self._handle, fkv.ctypes.data, B.es(fkv, 0), B.es(fkv, 1),
                                               fiv.ctypes.data, B.es(fiv, 0), sp, ssj, ssk, C.byref(its)))
        return int(its.value)
