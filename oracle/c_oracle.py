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
    return _lib


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
