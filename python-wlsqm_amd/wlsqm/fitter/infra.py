"""Import-compatibility placeholder for the reference's ``wlsqm.fitter.infra``.

In the reference this module holds the per-case bookkeeping (Case, CaseManager, Allocator, remap, number_of_dofs;
infra.pyx:67-880) as ``cdef`` structs and functions with no Python-callable names.  The GPU build has no per-case host
objects: the batch is described by the arrays themselves (``struct wlsqm_batch`` in include/wlsqm_hip.h).  The three pure
functions of that module that matter for parity are exported by the C ABI and re-exposed here.
"""
from .. import _binding as _B


def number_of_dofs(dimension, order):
    """infra.pyx:67-112 (also available as wlsqm.number_of_dofs)."""
    return int(_B.lib().wlsqm_hip_number_of_dofs(int(dimension), int(order)))


def number_of_reduced_dofs(n, mask):
    """infra.pyx:119-121: n minus the number of set bits of the knowns mask (bits >= n are NOT masked off)."""
    return int(_B.lib().wlsqm_hip_number_of_reduced_dofs(int(n), int(mask)))


def remap(n, mask):
    """infra.pyx:145-200: (nr, o2r, r2o) index maps of the knowns elimination, -1 sentinels included (bit-exact contract)."""
    import ctypes as C
    import numpy as np
    o2r = np.empty(int(n), np.int32); r2o = np.empty(int(n), np.int32)
    nr = _B.lib().wlsqm_hip_remap(o2r.ctypes.data_as(C.c_void_p), r2o.ctypes.data_as(C.c_void_p), int(n), int(mask))
    return int(nr), o2r, r2o
