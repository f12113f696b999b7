"""`libcloudphxx.common` of the reference's Python module (ref: bindings/python/common.hpp:19-172, lib.cpp:55-66,129-144):
constants and scalar formula functions, evaluated on the host by the product library's formula code
(`lcx_common_eval`, include/lcx.h) -- the same `lcx_math.hpp` expressions the kernels use.  Only what the Lagrangian path and
its tests use is provided (no chemistry constants, no ice)."""
import ctypes as C

from . import _lib

_fn = None


def _eval(name, *args, lib=None, prefix="lcx_"):
    global _fn
    if lib is None:
        if _fn is None:
            _fn = _lib.load().lcx_common_eval
        fn, err = _fn, _lib.load().lcx_last_error
    else:
        fn, err = getattr(lib, prefix + "common_eval"), getattr(lib, prefix + "last_error")
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
    a = (C.c_double * max(len(args), 1))(*[float(x) for x in args])
    out = C.c_double()
    if fn(name.encode(), a, len(args), C.byref(out)):
        err.restype = C.c_char_p
        raise RuntimeError(err().decode())
    return out.value


_CONSTANTS = ("R_d", "R_v", "c_pd", "c_pv", "c_pw", "g", "p_1000", "eps", "rho_stp", "rho_w")
_FUNCTIONS = {"th_dry2std": 2, "th_std2dry": 2, "exner": 1, "p_v": 2, "p_vs": 1, "r_vs": 2, "p_vs_tet": 1, "l_v": 1, "T": 2,
              "p": 3, "visc": 1, "rw3_cr": 3, "S_cr": 3, "p_hydro": 5, "rhod": 3}


def _make(name, nargs):
    def f(*args):
        if len(args) != nargs:
            raise TypeError("common.%s() takes %d arguments (%d given)" % (name, nargs, len(args)))
        return _eval(name, *args)
    f.__name__ = name
    return f


for _n in _CONSTANTS:
    globals()[_n] = _eval(_n)
for _n, _k in _FUNCTIONS.items():
    globals()[_n] = _make(_n, _k)
del _n, _k
