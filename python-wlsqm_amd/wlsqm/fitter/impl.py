"""Import-compatibility placeholder for the reference's ``wlsqm.fitter.impl``.

In the reference this module holds the arithmetic of the fit (make_c_*D, make_A, preprocess_A, solve, solve_iterative;
impl.pyx:70-1083) as ``cdef ... nogil`` functions: it has no Python-callable names, only a Cython-level API.  Here that
arithmetic lives in the HIP kernels (python-wlsqm_amd/csrc/wlsqm_kernels.hpp, wlsqm_moments.hpp, fit_*.hip) behind the C ABI
of include/wlsqm_hip.h; this module exists so that ``from wlsqm.fitter import impl`` keeps working.
"""
