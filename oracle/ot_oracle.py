"""oracle/ot_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end for ``oracle/ot_oracle.c`` (this repo's plain-C restatement of the
reference's ``ot_func.cpp``) plus a numpy restatement of the Python driver
(``/root/reference/SpaDOT/utils/OT_loss/ot_solvers.py``).  Parity status: PINNED by
``tests/golden/ot_*.npz`` (generated from the reference by ``oracle/gen_golden_ot.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Nothing under ``spadot_amd/`` does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C oracle (and, where /root/reference is mounted, oracle/_ref)."""
    so = os.path.join(_HERE, "liboracle_ot.so")
    src = os.path.join(_HERE, "ot_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_ot.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_update_process.restype = ctypes.c_double
        _LIB.orc_primal.restype = ctypes.c_double
        _LIB.orc_dual.restype = ctypes.c_double
        _LIB.orc_duality_gap.restype = ctypes.c_double
        _LIB.orc_transport_duality_gap.restype = ctypes.c_double
        _LIB.orc_step1.restype = ctypes.c_int
    return _LIB


def ref_lib():
    """oracle/_ref/libot_ref.so -- the reference's own ot_func.cpp compiled by oracle/Makefile.
    Returns None when it has not been built (it is optional everywhere)."""
    path = os.path.join(_HERE, "_ref", "libot_ref.so")
    if not os.path.exists(path):
        return None
    L = ctypes.CDLL(path)
    L.update_process_double.restype = ctypes.c_double
    L.compute_duality_gap_double.restype = ctypes.c_double
    L.primal_double.restype = ctypes.c_double
    L.dual_double.restype = ctypes.c_double
    L.step1_process_double.restype = ctypes.c_int
    return L


def _p(x):
    assert x.dtype == np.float64 and x.flags["C_CONTIGUOUS"]
    return x.ctypes.data_as(_D)


def _c(x):
    return ctypes.c_double(float(x))


# ---- thin wrappers, same argument order as the reference shim (ot_func.py:317-567) ----

def update_K(K, Kbar, C, u, v, eps):
    m, n = C.shape
    lib().orc_update_k(_p(K), _p(Kbar), _p(C), _p(u), _p(v), _c(eps), m, n)


def update_R(R, K, a, b):
    m, n = K.shape
    lib().orc_update_R(_p(R), _p(K), _p(a), _p(b), m, n)


def primal(C, Kbar, R, dx, dy, p, q, a, b, eps, l1, l2):
    m, n = C.shape
    return lib().orc_primal(_p(C), _p(Kbar), _p(R), _p(dx), _p(dy), _p(p), _p(q), _p(a), _p(b),
                            _c(eps), _c(l1), _c(l2), m, n)


def dual(C, Kbar, R, dx, dy, p, q, a, b, eps, l1, l2):
    m, n = C.shape
    return lib().orc_dual(_p(C), _p(Kbar), _p(R), _p(dx), _p(dy), _p(p), _p(q), _p(a), _p(b),
                          _c(eps), _c(l1), _c(l2), m, n)


def duality_gap(C, Kbar, R, dx, dy, p, q, a, b, eps, l1, l2):
    m, n = C.shape
    return lib().orc_duality_gap(_p(C), _p(Kbar), _p(R), _p(dx), _p(dy), _p(p), _p(q), _p(a),
                                 _p(b), _c(eps), _c(l1), _c(l2), m, n)


def update_a_b(a, b, K, dx, dy, p, q, u, v, l1, l2, al1, al2, eps):
    m, n = K.shape
    lib().orc_update_a_b(_p(a), _p(b), _p(K), _p(dx), _p(dy), _p(p), _p(q), _p(u), _p(v),
                         _c(l1), _c(l2), _c(al1), _c(al2), _c(eps), m, n)


def step1_process(a, b, old_a, old_b, K, C, dx, dy, p, q, u, v, cur_iter, max_iter, iters, tau,
                  l1, l2, al1, al2, eps):
    m, n = K.shape
    return lib().orc_step1(_p(a), _p(b), _p(old_a), _p(old_b), _p(K), _p(C), _p(dx), _p(dy),
                           _p(p), _p(q), _p(u), _p(v), int(cur_iter), int(max_iter), int(iters),
                           _c(tau), _c(l1), _c(l2), _c(al1), _c(al2), _c(eps), m, n)


def update_process(R, a, b, old_a, old_b, K, Kbar, C, dx, dy, p, q, u, v, eps_scalings,
                   cur_scaling, batch_size, eps, threshold, tau, l1, l2, al1, al2, cur_iter,
                   max_iter):
    """Returns (gap, scaling iterations run)."""
    m, n = K.shape
    it = ctypes.c_int(0)
    gap = lib().orc_update_process(
        _p(R), _p(a), _p(b), _p(old_a), _p(old_b), _p(K), _p(Kbar), _p(C), _p(dx), _p(dy), _p(p),
        _p(q), _p(u), _p(v), int(eps_scalings), int(cur_scaling), int(batch_size), _c(eps),
        _c(threshold), _c(tau), _c(l1), _c(l2), _c(al1), _c(al2), int(cur_iter), int(max_iter),
        m, n, ctypes.byref(it))
    return gap, it.value


# ---- driver restatement: ot_solvers.py:164-449 and :95-121 ----

def optimal_transport_duality_gap(C, G, lambda1, lambda2, epsilon, batch_size, tolerance, tau,
                                  epsilon0, max_iter, return_info=False, **ignored):
    """Whole solve in C (orc_transport_duality_gap).  Returns R/J like ot_solvers.py:449 and
    raises on a NaN gap like ot_solvers.py:446-447."""
    C = np.ascontiguousarray(C, dtype=np.float64)
    G = np.ascontiguousarray(G, dtype=np.float64)
    I, J = C.shape
    plan = np.empty((I, J), dtype=np.float64)
    iters = np.zeros(6, dtype=np.int32)
    u = np.empty(I)
    v = np.empty(J)
    gap = lib().orc_transport_duality_gap(
        _p(plan), _p(C), _p(G), I, J, _c(lambda1), _c(lambda2), _c(epsilon), int(batch_size),
        _c(tolerance), _c(tau), _c(epsilon0), int(max_iter), iters.ctypes.data_as(_I), _p(u), _p(v))
    if np.isnan(gap):
        raise RuntimeError("Overflow encountered in duality gap computation, please report this incident")
    if return_info:
        return plan, {"gap": gap, "stage_iters": iters, "u": u, "v": v}
    return plan


def sqeuclidean_cost(a, b):
    """sklearn pairwise_distances(metric='sqeuclidean') arithmetic: |a|^2 + |b|^2 - 2ab^T,
    clipped at 0 (ot_solvers.py:102)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = -2.0 * (a @ b.T)
    d += (a * a).sum(axis=1)[:, None]
    d += (b * b).sum(axis=1)[None, :]
    np.maximum(d, 0, out=d)
    return d


def compute_transport_map(a, b, config, C=None, G=None, all_growth_iters=True):
    """ot_solvers.py:95-121.  Cost = sqeuclidean / median; `growth_iters` solves feeding row
    sums back as growth; the FIRST solve is what is returned (quirk kept: SURVEY App. D.1).
    With all_growth_iters=False the discarded solves 2..n are skipped (same return value)."""
    if C is None:
        C = sqeuclidean_cost(a, b)
        C = C / np.median(C)
    cfg = dict(config)
    g = np.ones(C.shape[0]) if G is None else np.asarray(G, dtype=np.float64)
    first = None
    n_solves = int(cfg["growth_iters"]) if all_growth_iters else 1
    gamma = None
    for i in range(n_solves):
        row_sums = g if i == 0 else gamma.sum(axis=1)
        kw = {k: cfg[k] for k in ("lambda1", "lambda2", "epsilon", "batch_size", "tolerance",
                                  "tau", "epsilon0", "max_iter")}
        gamma = optimal_transport_duality_gap(C, row_sums, **kw)
        if first is None:
            first = gamma.copy()
    return first
