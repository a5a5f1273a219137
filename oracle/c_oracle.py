"""ctypes binding of oracle/libgvi_oracle.so (test infrastructure / cpu_baseline only)."""
import ctypes as C
import os

import numpy as np

import build_oracle

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle.build())
        _lib.gvi_oracle_moments.restype = C.c_int
        _lib.gvi_oracle_moments.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long] + [C.c_void_p] * 4 + \
            [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3
        _lib.gvi_oracle_max_threads.restype = C.c_int
        _lib.gvi_oracle_set_sdf2d.restype = C.c_int
        _lib.gvi_oracle_set_sdf2d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    return _lib


_sdf_keep = None


def set_sdf2d(origin, cell, field):
    """Grid of the HINGE_SDF_2D kind (kind 4): field[r, c] at origin + (c, r) cell, handed over column-major as the
    reference stores it (helpers/CudaOperation.h:130)."""
    global _sdf_keep
    f = np.asfortranarray(np.asarray(field, dtype=np.float64))
    _sdf_keep = f                                       # the library keeps the pointer
    lib().gvi_oracle_set_sdf2d(f.ctypes.data_as(C.c_void_p), f.shape[0], f.shape[1], float(origin[0]), float(origin[1]), float(cell))


def max_threads():
    return lib().gvi_oracle_max_threads()


def moments(Z, w, mu, Sigma, kind, params, n, temperature=None, fused=False, nthreads=0):
    """Reference-shaped (fused=False) or single-pass (fused=True) CPU moments of K factors."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    Z, w, mu, Sigma, params = f(Z), f(w), f(mu), f(Sigma), f(params)
    K, d = mu.shape
    params = params.reshape(K, -1)
    temp = f(np.ones(K) if temperature is None else temperature)
    Ephi, Vdmu, Vddmu = np.empty(K), np.empty((K, d)), np.empty((K, d, d))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().gvi_oracle_moments(K, d, n, Z.shape[0], p(Z), p(w), p(mu), p(Sigma), kind, p(params), params.shape[1],
                             p(temp), int(fused), nthreads, p(Ephi), p(Vdmu), p(Vddmu))
    return Ephi, Vdmu, Vddmu
