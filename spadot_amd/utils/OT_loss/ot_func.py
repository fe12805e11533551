"""Mirror of the reference ctypes shim (/root/reference/SpaDOT/utils/OT_loss/ot_func.py) bound to
libspadot_ot.so instead of libot.so: same wrapper names, argument order and in-place behaviour
(`*_c` functions at ot_func.py:317-567).  Arrays are host numpy buffers, exactly as in the
reference; every call runs on the MI355X (upload -> HIP kernels -> write-back).

The hot path does not go through here -- it uses the device-resident solver (spadot_amd.ot) --
but anything written against the reference shim keeps working, which is what makes the library
a drop-in for libot.so.
"""
import ctypes

import numpy as np
from numpy.ctypeslib import ndpointer

from ..._lib import _load

lib = _load("libspadot_ot.so")

_d1 = ndpointer(dtype=ctypes.c_double, ndim=1, flags="C_CONTIGUOUS")
_d2 = ndpointer(dtype=ctypes.c_double, ndim=2, flags="C_CONTIGUOUS")
_f1 = ndpointer(dtype=ctypes.c_float, ndim=1, flags="C_CONTIGUOUS")
_f2 = ndpointer(dtype=ctypes.c_float, ndim=2, flags="C_CONTIGUOUS")
_cd, _cf, _ci = ctypes.c_double, ctypes.c_float, ctypes.c_int

for _name in ("dummy", "primal", "dual", "compute_duality_gap"):
    getattr(lib, _name + "_double").argtypes = [_d2, _d2, _d2, _d1, _d1, _d1, _d1, _d1, _d1, _cd, _cd, _cd, _ci, _ci]
    getattr(lib, _name + "_double").restype = _cd
    getattr(lib, _name + "_float").argtypes = [_f2, _f2, _f2, _f1, _f1, _f1, _f1, _f1, _f1, _cf, _cf, _cf, _ci, _ci]
    getattr(lib, _name + "_float").restype = _cf
lib.update_k_double.argtypes = [_d2, _d2, _d2, _d1, _d1, _cd, _ci, _ci]
lib.update_k_double.restype = None
lib.update_k_float.argtypes = [_f2, _f2, _f2, _f1, _f1, _cf, _ci, _ci]
lib.update_k_float.restype = None
lib.update_R_double.argtypes = [_d2, _d2, _d1, _d1, _ci, _ci]
lib.update_R_double.restype = None
lib.update_R_float.argtypes = [_f2, _f2, _f1, _f1, _ci, _ci]
lib.update_R_float.restype = None
lib.step1_process_double.argtypes = [_d1, _d1, _d1, _d1, _d2, _d2, _d1, _d1, _d1, _d1, _d1, _d1,
                                     _ci, _ci, _ci, _cd, _cd, _cd, _cd, _cd, _cd, _ci, _ci]
lib.step1_process_double.restype = _ci
# the reference lists 26 argtypes and passes 28 values (ot_func.py:286-313 vs :561-567); the full
# 28-entry C signature is declared here
lib.update_process_double.argtypes = [_d2, _d1, _d1, _d1, _d1, _d2, _d2, _d2, _d1, _d1, _d1, _d1, _d1, _d1,
                                      _ci, _ci, _ci, _cd, _cd, _cd, _cd, _cd, _cd, _cd, _ci, _ci, _ci, _ci]
lib.update_process_double.restype = _cd


def _as(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


def _gap_like(name, C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float):
    m, n = C.shape
    if use_float:
        f = ctypes.c_float
        return getattr(lib, name + "_float")(_as(C, f), _as(K, f), _as(R, f), _as(dx, f), _as(dy, f), _as(p, f),
                                             _as(q, f), _as(a, f), _as(b, f), epsilon, lambda1, lambda2, m, n)
    return getattr(lib, name + "_double")(C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, m, n)


def dummy_c(C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float=False):
    return _gap_like("dummy", C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float)


def primal_c(C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float=False):
    return _gap_like("primal", C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float)


def dual_c(C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float=False):
    return _gap_like("dual", C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float)


def compute_duality_gap_c(C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float=False):
    return _gap_like("compute_duality_gap", C, K, R, dx, dy, p, q, a, b, epsilon, lambda1, lambda2, use_float)


def update_K_c(K, _K, C, u, v, epsilon, use_float=False):
    m, n = C.shape
    (lib.update_k_float if use_float else lib.update_k_double)(K, _K, C, u, v, epsilon, m, n)


def update_R_c(R, K, a, b, use_float=False):
    m, n = K.shape
    (lib.update_R_float if use_float else lib.update_R_double)(R, K, a, b, m, n)


def update_a_b_c(*args, **kwargs):
    # ot_func.py:479-519 calls update_a_b_{float,double}, which libot.so does not export
    # (argtypes block commented out, ot_func.py:212-256): calling it raises AttributeError there.
    raise AttributeError("update_a_b_double is not exported by the reference libot.so either")


def step1_process_c(a, b, old_a, old_b, K, C, dx, dy, p, q, u, v,
                    cur_iter, max_iter, iters, tau, lambda1, lambda2, alpha1, alpha2, epsilon):
    m, n = K.shape
    return lib.step1_process_double(a, b, old_a, old_b, K, C, dx, dy, p, q, u, v, cur_iter, int(max_iter),
                                    iters, float(tau), lambda1, lambda2, alpha1, alpha2, epsilon, m, n)


def update_process_c(R, a, b, old_a, old_b, K, _K, C, dx, dy, p, q, u, v,
                     epsilon_scaling, cur_epsilon_scaling, batch_size, epsilon, threshold,
                     tau, lambda1, lambda2, alpha1, alpha2, cur_iter, max_iter):
    m, n = K.shape
    return lib.update_process_double(R, a, b, old_a, old_b, K, _K, C, dx, dy, p, q, u, v,
                                     epsilon_scaling, cur_epsilon_scaling, batch_size, epsilon, threshold,
                                     float(tau), lambda1, lambda2, alpha1, alpha2, cur_iter, int(max_iter), m, n)
