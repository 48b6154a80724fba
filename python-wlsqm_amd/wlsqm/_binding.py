"""ctypes binding of libwlsqm_hip.so (C ABI declared in include/wlsqm_hip.h).

There is no CPU fallback: if the shared library is missing the import of any fitting
entry point fails loudly, and if no HIP device is present the library itself returns
WLSQM_ENODEVICE (raised here as RuntimeError).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libwlsqm_hip.so")

WLSQM_OK, WLSQM_EVALUE, WLSQM_ERUNTIME, WLSQM_EMEMORY, WLSQM_ENODEVICE = 0, -1, -2, -3, -4


class Batch(C.Structure):
    """struct wlsqm_batch (include/wlsqm_hip.h)."""
    _fields_ = [
        ("dimension", C.c_int32), ("do_sens", C.c_int32), ("ncases", C.c_int64),
        ("xk", C.c_void_p), ("xk_stride_case", C.c_int64), ("xk_stride_k", C.c_int64),
        ("fk", C.c_void_p), ("fk_stride_case", C.c_int64), ("fk_stride_k", C.c_int64),
        ("nk", C.c_void_p), ("nk_stride", C.c_int64),
        ("xi", C.c_void_p), ("xi_stride_case", C.c_int64),
        ("fi", C.c_void_p), ("fi_stride_case", C.c_int64),
        ("sens", C.c_void_p), ("sens_stride_case", C.c_int64), ("sens_stride_k", C.c_int64),
        ("order", C.c_void_p), ("order_stride", C.c_int64),
        ("knowns", C.c_void_p), ("knowns_stride", C.c_int64),
        ("weighting_method", C.c_void_p), ("wm_stride", C.c_int64),
        ("iterative", C.c_int32), ("max_iter", C.c_int32), ("max_nk", C.c_int64),
    ]


_lib = None


def _init_torch_hip_first():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, no SONAME) next to the
    system one this library links (libamdhip64.so.7).  Both can live in one process, but only if torch's is
    initialised FIRST (the other order makes torch report "No HIP GPUs are available").  torch is the designated
    owner of device memory and streams for the device-resident API, so when it is installed we let it
    initialise the GPU before libwlsqm_hip.so is loaded.  Set WLSQM_HIP_SKIP_TORCH_INIT=1 to skip."""
    if os.environ.get("WLSQM_HIP_SKIP_TORCH_INIT") == "1":
        return
    try:
        import torch
    except Exception:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def lib():
    """Load libwlsqm_hip.so; raises ImportError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "wlsqm: the HIP library %s is missing. Build it with python-wlsqm_amd/build.sh "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _init_torch_hip_first()
    L = C.CDLL(LIB_PATH)
    L.wlsqm_hip_last_error.restype = C.c_char_p
    L.wlsqm_hip_last_kernel.restype = C.c_char_p
    L.wlsqm_hip_device_count.restype = C.c_int
    L.wlsqm_hip_number_of_dofs.argtypes = [C.c_int, C.c_int]
    L.wlsqm_hip_number_of_reduced_dofs.argtypes = [C.c_int, C.c_int64]
    L.wlsqm_hip_remap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
    L.wlsqm_hip_fit_many_host.argtypes = [C.POINTER(Batch), C.c_int, C.POINTER(C.c_int32)]
    L.wlsqm_hip_fit_many_device.argtypes = [C.POINTER(Batch), C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int64, C.POINTER(C.c_int32)]
    L.wlsqm_hip_fit_many_device_orders.argtypes = [C.POINTER(Batch), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int32)]
    L.wlsqm_hip_fit_many_device_orders.restype = C.c_int
    L.wlsqm_hip_set_strict.argtypes = [C.c_int]
    L.wlsqm_hip_set_strict.restype = C.c_int
    L.wlsqm_hip_get_strict.restype = C.c_int
    if hasattr(L, "wlsqm_hip_set_row_hint"):                         # (tools/ab_lib.sh times older builds of the library through this binding)
        L.wlsqm_hip_set_row_hint.argtypes = [C.c_int]
        L.wlsqm_hip_set_row_hint.restype = C.c_int
        L.wlsqm_hip_set_order_hint.argtypes = [C.c_int]
        L.wlsqm_hip_set_order_hint.restype = C.c_int
    L.wlsqm_hip_strict_intermediates_device.argtypes = [C.POINTER(Batch), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_int64]
    L.wlsqm_hip_strict_intermediates_device.restype = C.c_int
    L.wlsqm_hip_time_fit_device.argtypes = [C.POINTER(Batch), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                            C.POINTER(C.c_float)]
    cloud = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.wlsqm_hip_fit_cloud_device.argtypes = cloud + [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_void_p, C.POINTER(C.c_int32)]
    L.wlsqm_hip_time_fit_cloud_device.argtypes = cloud + [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.wlsqm_hip_expert_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.wlsqm_hip_knn_device.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.wlsqm_hip_knn_subset_device.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    L.wlsqm_hip_ball_device.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p]
    L.wlsqm_hip_nearest_device.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                           C.c_void_p]
    L.wlsqm_hip_expert_interpolate_nearest.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                                       C.c_void_p]
    L.wlsqm_hip_expert_interpolate_continuous.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_int,
                                                          C.c_void_p]
    L.wlsqm_hip_expert_create_guest.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.wlsqm_hip_expert_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    L.wlsqm_hip_expert_prepare_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]
    L.wlsqm_hip_expert_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]
    L.wlsqm_hip_expert_solve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    L.wlsqm_hip_expert_prepare_operator.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.wlsqm_hip_expert_solve_many_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                                     C.c_void_p, C.c_int64, C.c_int64]
    L.wlsqm_hip_expert_solve_many.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                              C.c_void_p, C.c_int64, C.c_int64]
    L.wlsqm_hip_expert_memory_used.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.wlsqm_hip_expert_destroy.argtypes = [C.c_void_p]
    L.wlsqm_hip_expert_conds.argtypes = [C.c_void_p, C.c_void_p]
    L.wlsqm_hip_expert_interpolate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_double, C.c_int, C.c_void_p]
    L.wlsqm_hip_interpolate_fit_host.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.c_int64, C.c_int, C.c_void_p, C.c_int]
    for name in ("wlsqm_hip_fit_many_host", "wlsqm_hip_fit_many_device", "wlsqm_hip_time_fit_device",
                 "wlsqm_hip_expert_create", "wlsqm_hip_expert_create_guest", "wlsqm_hip_expert_prepare", "wlsqm_hip_expert_prepare_device", "wlsqm_hip_expert_solve",
                 "wlsqm_hip_expert_solve_device", "wlsqm_hip_expert_prepare_operator", "wlsqm_hip_expert_solve_many_device", "wlsqm_hip_expert_solve_many",
                 "wlsqm_hip_expert_memory_used", "wlsqm_hip_expert_destroy",
                 "wlsqm_hip_expert_conds", "wlsqm_hip_expert_interpolate", "wlsqm_hip_interpolate_fit_host",
                 "wlsqm_hip_fit_cloud_device", "wlsqm_hip_time_fit_cloud_device", "wlsqm_hip_knn_device", "wlsqm_hip_knn_subset_device", "wlsqm_hip_ball_device", "wlsqm_hip_nearest_device",
                 "wlsqm_hip_expert_interpolate_nearest", "wlsqm_hip_expert_interpolate_continuous",
                 "wlsqm_hip_number_of_dofs", "wlsqm_hip_number_of_reduced_dofs", "wlsqm_hip_remap"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc):
    """Map a WLSQM_E* code to the exception class the reference raises (SURVEY §8b 'Errors')."""
    if rc == WLSQM_OK:
        return
    msg = lib().wlsqm_hip_last_error().decode("utf-8", "replace")
    if rc == WLSQM_EVALUE:
        raise ValueError(msg)
    if rc == WLSQM_EMEMORY:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def default_device():
    """Device of the host-array entry points, in this order: WLSQM_HIP_DEVICE if set; torch's current device when torch has
    initialised the GPU in this process AND that device is not the default 0 (the caller said torch.cuda.set_device /
    torch.cuda.device); LOCAL_RANK when set (one process per GPU under torch.distributed.run: a rank that only ever named
    'cuda:<local_rank>' explicitly still has current_device() == 0) AND that ordinal exists — a launcher that isolates one GPU per
    rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES) leaves every rank with device 0 only, and LOCAL_RANK = 3 names nothing there;
    otherwise torch's current device, otherwise 0."""
    if "WLSQM_HIP_DEVICE" in os.environ:
        return int(os.environ["WLSQM_HIP_DEVICE"])
    import sys
    torch = sys.modules.get("torch")
    cur = None
    if torch is not None:
        try:
            if torch.cuda.is_initialized():
                cur = int(torch.cuda.current_device())
        except Exception:
            cur = None
    if cur:
        return cur
    if "LOCAL_RANK" in os.environ:
        lr = int(os.environ["LOCAL_RANK"])
        try:
            ndev = int(lib().wlsqm_hip_device_count())
        except Exception:
            ndev = 0
        if 0 <= lr < ndev or ndev <= 0:                 # (no visible device at all: keep the rank's number, the call fails loudly anyway)
            return lr
    return cur or 0


# ---- argument coercion with the reference's typed-memoryview rules (SURVEY §8b 'Array typing') ----

def view(a, dtype, ndim, name, contiguous_last=False, writable=False):
    """np.ndarray view of `a` with Cython-memoryview-like checks: exact dtype, exact ndim, and
    (when the reference declares ::view.contiguous) unit stride on the last axis."""
    arr = np.asarray(a)
    if arr.dtype != np.dtype(dtype):
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s' (argument %s)"
                         % (np.dtype(dtype).name, arr.dtype.name, name))
    if arr.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d) (argument %s)"
                         % (ndim, arr.ndim, name))
    if contiguous_last and arr.shape[-1] > 1 and arr.strides[-1] != arr.itemsize:
        raise ValueError("Buffer and memoryview are not contiguous in the same dimension. (argument %s)" % name)
    if any(s % arr.itemsize for s in arr.strides):
        raise ValueError("argument %s: strides must be multiples of the item size" % name)
    if writable and not arr.flags.writeable:
        raise ValueError("buffer source array is read-only (argument %s)" % name)
    return arr


def es(arr, axis):
    """element stride"""
    return arr.strides[axis] // arr.itemsize
