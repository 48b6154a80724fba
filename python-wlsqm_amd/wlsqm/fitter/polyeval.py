"""Import-compatibility placeholder for the reference's ``wlsqm.fitter.polyeval``.

In the reference this module evaluates the fitted Taylor / general polynomials (taylor_{1,2,3}D, general_{1,2,3}D;
polyeval.pyx:82-951) as ``cdef ... nogil`` functions with no Python-callable names; the Python-level entry points are
``wlsqm.interpolate_fit`` / ``lambdify_fit`` and ``ExpertSolver.interpolate``.  Here the evaluation runs on the GPU
(python-wlsqm_amd/csrc/wlsqm_interp.hpp, interp.hip) behind the same entry points.
"""
