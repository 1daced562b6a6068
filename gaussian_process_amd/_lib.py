"""ctypes binding of libgpx.so (include/gpx.h).  There is no CPU fallback: if the HIP library
is missing or no MI355X is present, every compute entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPX_LIB_PATH", os.path.join(_HERE, "libgpx.so"))  # env: development override

GPX_KIND = {"rbf": 0, "ard": 1}
GPX_FULL, GPX_LOWER = 0, 1
GPX_NB = 128

_dp = C.c_void_p      # device pointer
_hp = C.POINTER(C.c_double)  # host double*
_i, _d, _sz = C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); one entry per symbol declared in include/gpx.h
SIGNATURES = {
    "gpx_create": (_i, [C.POINTER(C.c_void_p), _i]),
    "gpx_destroy": (_i, [C.c_void_p]),
    "gpx_set_stream": (_i, [C.c_void_p, C.c_void_p]),
    "gpx_synchronize": (_i, [C.c_void_p]),
    "gpx_last_error": (C.c_char_p, []),
    "gpx_version": (C.c_char_p, []),
    "gpx_dinv_elems": (_sz, [_i]),
    "gpx_gram": (_i, [C.c_void_p, _i, _dp, _i, _i, _dp, _i, _i, _i, _d, _hp, _d, _dp, _i, _i]),
    "gpx_gram_dtheta": (_i, [C.c_void_p, _i, _dp, _i, _i, _dp, _i, _i, _i, _d, _hp, _dp, _dp, _dp, _i, _sz]),
    "gpx_potrf": (_i, [C.c_void_p, _i, _dp, _i, _dp]),
    "gpx_potrf_async": (_i, [C.c_void_p, _i, _dp, _i, _dp, _dp, _i]),
    "gpx_trsm_right": (_i, [C.c_void_p, _i, _i, _dp, _i, _dp, _dp, _i]),
    "gpx_diag_inv": (_i, [C.c_void_p, _i, _dp, _i, _dp]),
    "gpx_trsm": (_i, [C.c_void_p, _i, _i, _i, _dp, _i, _dp, _dp, _i]),
    "gpx_logdet": (_i, [C.c_void_p, _i, _dp, _i, _hp]),
    "gpx_trtri": (_i, [C.c_void_p, _i, _dp, _i, _dp]),
    "gpx_lauum": (_i, [C.c_void_p, _i, _dp, _i, _dp, _i]),
    "gpx_lauum_upper": (_i, [C.c_void_p, _i, _dp, _i, _dp]),
    "gpx_syrk_sub": (_i, [C.c_void_p, _i, _i, _dp, _i, _dp, _i, _d, _i]),
    "gpx_gemm": (_i, [C.c_void_p, _i, _i, _i, _i, _i, _d, _dp, _i, _dp, _i, _d, _dp, _i]),
    "gpx_gemm_colblocks": (_i, [C.c_void_p, _i, _i, _i, _d, _dp, _i, _dp, _i, _d, _dp, _i, _i, _dp, _dp]),
    "gpx_trmv": (_i, [C.c_void_p, _i, _i, _dp, _i, _dp, _dp]),
    "gpx_trmm": (_i, [C.c_void_p, _i, _i, _i, _dp, _i, _dp, _i, _dp, _i]),
    "gpx_trmm_right": (_i, [C.c_void_p, _i, _i, _dp, _i, _dp, _i, _dp, _i]),
    "gpx_gemv": (_i, [C.c_void_p, _i, _i, _i, _dp, _i, _dp, _dp]),
    "gpx_gemv_acc": (_i, [C.c_void_p, _i, _i, _i, _d, _dp, _i, _dp, _d, _dp]),
    "gpx_colsumsq_sub": (_i, [C.c_void_p, _i, _i, _dp, _i, _d, _dp]),
    "gpx_gemv_static": (_i, [C.c_void_p, _i, _i, _i, _dp, _i, _dp, _dp]),
    "gpx_dot": (_i, [C.c_void_p, _i, _dp, _dp, _hp]),
    "gpx_symmetrize": (_i, [C.c_void_p, _i, _dp, _i]),
    "gpx_diag": (_i, [C.c_void_p, _i, _dp, _i, _dp]),
    "gpx_grad_trace": (_i, [C.c_void_p, _i, _dp, _i, _i, _i, _d, _hp, _dp, _dp, _i, _dp, _hp]),
    "gpx_grad_trace_slab": (_i, [C.c_void_p, _i, _dp, _i, _i, _i, _d, _hp, _dp, _dp, _i, _i, _i, _i, _i, _hp]),
    "gpx_gpr_lml_grad": (_i, [C.c_void_p, _i, _dp, _i, _i, _i, _d, _hp, _d, _dp, _dp, _i, _dp, _dp,
                              _dp, _dp, _hp, _hp]),
    "gpx_gpr_grad_from_factor": (_i, [C.c_void_p, _i, _dp, _i, _i, _i, _d, _hp, _dp, _i, _dp, _dp, _dp, _hp]),
    "gpx_gpr_predict": (_i, [C.c_void_p, _i, _dp, _i, _i, _i, _d, _hp, _d, _dp, _dp, _i, _dp, _dp, _i, _i,
                             _dp, _i, _dp, _dp, _dp, _i, _dp, _i]),
    "gpx_gpr_predict_inv": (_i, [C.c_void_p, _i, _dp, _i, _i, _i, _d, _hp, _d, _dp, _dp, _i, _dp, _i, _i, _dp, _dp,
                                 _i, _dp, _dp, _i, _dp, _i]),
    "gpx_ep_scale": (_i, [C.c_void_p, _i, _dp, _i, _dp, _dp, _i, _dp, _i]),
    "gpx_ep_sweep": (_i, [C.c_void_p, _i, _dp, _i, _dp, _dp, _dp, C.POINTER(C.c_int), _i, _dp]),
    "gpx_ep_sweep_work_elems": (C.c_size_t, [_i]),
    "gpx_ep_recompute": (_i, [C.c_void_p, _i, _dp, _i, _dp, _dp, _dp, _dp, _dp, _i, _dp, _dp, _dp]),
}

_lib = None


class GpxError(RuntimeError):
    pass


def load():
    """Load libgpx.so and set the prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpxError(
            "libgpx.so not found at %s: build it with `python -m gaussian_process_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().gpx_last_error().decode()


def check(rc, what):
    """Map a gpx return code: <0 -> GpxError, >0 -> numpy LinAlgError (scipy's wording,
    regression.py:36,49), 0 -> ok."""
    if rc == 0:
        return
    if rc > 0:
        raise np.linalg.LinAlgError(
            "%d-th leading minor of the array is not positive definite" % rc)
    raise GpxError("%s failed (%d): %s" % (what, rc, last_error()))


def host_doubles(values):
    arr = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
    return arr, arr.ctypes.data_as(_hp)
