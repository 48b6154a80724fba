"""wlsqm for AMD Instinct MI355X: the batched WLSQM (weighted least squares meshless) fit path of
Technologicat/python-wlsqm behind the reference's own Python API, running on hand-written HIP kernels.

Like the reference's wlsqm/__init__.py:25-28, the public names of the submodules are re-exported flat,
so they are available as ``wlsqm.fit_2D_many_parallel(...)``, ``wlsqm.ExpertSolver``, ``wlsqm.b2_F``:

    wlsqm.fitter.defs    # named constants (algorithms, weightings, DOF indices, bitmasks)
    wlsqm.fitter.simple  # simple fit API
    wlsqm.fitter.interp  # evaluation of fitted models (interpolate_fit, lambdify_fit)
    wlsqm.fitter.expert  # advanced API (prepare once / solve many, interpolate)
    wlsqm.hip            # device-resident entry points (torch tensors / raw pointers), bench hooks
"""
from pathlib import Path as _Path
__version__ = (_Path(__file__).parent / "VERSION").read_text().strip()

from .fitter.defs import *    # noqa: F401, F403
from .fitter.simple import *  # noqa: F401, F403
from .fitter.interp import *  # noqa: F401, F403
from .fitter.expert import *  # noqa: F401, F403
