"""ctypes binding of libtorchpiv_hip.so (include/torchpiv_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of this module raises, so a GPU box can never silently run without the HIP path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TPIV_LIB: development override (e.g. the stamped diagnostic build); the default is the in-tree library
LIB_PATH = os.environ.get("TPIV_LIB") or os.path.join(_HERE, "libtorchpiv_hip.so")

OK, EINVAL, EKEY, EHIP, ENOMEM, EUNSUPPORTED = 0, 1, 2, 3, 4, 5
MODE_DWS, MODE_CWS, MODE_CWS_FAST = 1, 2, 3
MODES = {"DWS": MODE_DWS, "CWS": MODE_CWS}           # the multipass modes of OfflinePIV (IterModMap, B:814-818)
ITER_MODES = dict(MODES, CWS_Fast=MODE_CWS_FAST)     # + piv_iteration_CWS_Fast (function-level seam only)
PREC_FAST, PREC_REFERENCE, PREC_F64, PREC_EXACT = 0, 1, 2, 3
PRECISIONS = {"fast": PREC_FAST, "reference": PREC_REFERENCE, "f64": PREC_F64, "exact": PREC_EXACT}
ABI_VERSION = 2


class HipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C torchpiv_amd/csrc)")
    return C.CDLL(LIB_PATH)


lib = _load()

_u8p = C.c_void_p
_f64p = C.c_void_p
_f32p = C.c_void_p
_int = C.c_int
_dbl = C.c_double
_vp = C.c_void_p

# every symbol include/torchpiv_hip.h declares
SIGNATURES = {
    "tpiv_version": (C.c_int, []),
    "tpiv_last_error": (C.c_char_p, []),
    "tpiv_field_shape": (C.c_int, [_int, _int, _int, _int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tpiv_coordinates": (C.c_int, [_int, _int, _int, _int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tpiv_spline_matrix": (C.c_int, [_int, C.POINTER(C.c_double), _int, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
    "tpiv_pass1": (C.c_int, [_u8p, _u8p, _int, _int, _int, _int, _int, _dbl, _int, _int, _f64p, _f64p, _u8p,
                             _vp, C.c_size_t, _vp]),
    "tpiv_work_bytes": (C.c_size_t, [_int, _int, _int, _int, _int]),
    "tpiv_predict": (C.c_int, [_int, _int, _int, _int, _int, _int, _f64p, _f64p, _f64p, _f64p, _u8p,
                               _f64p, _f64p, _f64p, _f64p, _f64p, _vp]),
    "tpiv_iter": (C.c_int, [_int, _u8p, _u8p, _int, _int, _int, _int, _int, _f64p, _f64p, _f64p, _f64p,
                            _dbl, _int, _int, _f64p, _f64p, _u8p, _f64p, _f64p, _vp, C.c_size_t, _vp]),
    "tpiv_plan_create": (C.c_int, [C.POINTER(C.c_void_p), _int, _int, _int, _int, _int, _int, _dbl, _dbl,
                                   _int, _int, _int]),
    "tpiv_plan_kernel_name": (C.c_char_p, [C.c_void_p, _int, C.c_char_p, _int]),
    "tpiv_plan_exact_fallbacks": (_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "tpiv_plan_exact_timing": (_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "tpiv_plan_destroy": (None, [C.c_void_p]),
    "tpiv_plan_n_pass": (C.c_int, [C.c_void_p]),
    "tpiv_plan_pass_geometry": (C.c_int, [C.c_void_p, _int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tpiv_plan_run": (C.c_int, [C.c_void_p, _u8p, _u8p, _int, _f64p, _f64p, _u8p, _vp]),
    "tpiv_plan_pass_fields": (C.c_int, [C.c_void_p, _int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p)]),
    "tpiv_postval": (C.c_int, [_f64p, _f64p, _u8p, _int, _int, _int, _u8p, _vp, _vp]),
    "tpiv_postval_compact": (C.c_int, [_f64p, _f64p, _u8p, _vp, _int, _int, _int, _vp, _vp, _f64p, _vp, _vp]),
    "tpiv_finish_fields": (C.c_int, [_f64p, _f64p, _int, _int, _int, C.c_double, C.c_double, _f64p, _f64p, _vp]),
    "tpiv_ensemble_moments": (C.c_int, [_f64p, _f64p, _int, C.c_longlong, _f64p, _vp]),
    "tpiv_bmp_unpack": (C.c_int, [_u8p, _vp, _u8p, _int, _int, _int, _u8p, _vp]),
    "tpiv_read_files": (C.c_int, [C.POINTER(C.c_char_p), _int, C.c_void_p, C.c_size_t, _int, C.POINTER(C.c_longlong)]),
    "tpiv_reader_open": (C.c_void_p, [C.POINTER(C.c_char_p), C.c_longlong, _int, C.POINTER(C.c_void_p), _int, C.c_size_t, _int]),
    "tpiv_reader_next": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "tpiv_reader_release": (C.c_int, [C.c_void_p]),
    "tpiv_reader_close": (None, [C.c_void_p]),
    "tpiv_plan_set_timing": (C.c_int, [C.c_void_p, _int]),
    "tpiv_plan_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), _int, C.POINTER(C.c_int)]),
    "tpiv_plan_debug_predict": (C.c_int, [C.c_void_p, _int, _int, _f64p, _f64p, _u8p, _f64p, _f64p, _f64p,
                                          _f64p, _vp]),
    "tpiv_debug_peaks": (C.c_int, [_f32p, _int, _int, _int, _dbl, _int, _f64p, _f64p, _u8p, _vp, C.c_size_t, _vp]),
    "tpiv_debug_pass": (C.c_int, [_int, _int, _u8p, _u8p, _int, _int, _int, _int, _int, _f64p, _f64p, _f64p, _f64p,
                                  _f64p, _u8p, _f32p, _f32p, _vp, C.c_size_t, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)        # AttributeError if the library lacks a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


if lib.tpiv_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} has ABI version {lib.tpiv_version()}, this package needs {ABI_VERSION}: rebuild "
                      "(make -C torchpiv_amd/csrc)")


def check(rc: int) -> None:
    """Map a status code to the exception type the reference raises in that situation."""
    if rc == OK:
        return
    msg = lib.tpiv_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == EKEY:
        raise KeyError(msg)
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise HipError(msg)
