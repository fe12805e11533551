"""CPU: the C/numpy oracle against the reference's own outputs (tests/golden/ot_*.npz,
made by oracle/gen_golden_ot.py) and, where it was built, against oracle/_ref/libot_ref.so
(the reference's ot_func.cpp compiled as-is)."""
import ctypes

import numpy as np
import pytest

from conftest import SOLVE_CASES, load_golden, solve_cfg


def _cost(oracle_ot, g):
    if g["C"].size:
        return np.ascontiguousarray(g["C"])
    C = oracle_ot.sqeuclidean_cost(g["a"], g["b"])
    return C / np.median(C)


@pytest.mark.parametrize("case", SOLVE_CASES)
def test_whole_solve_matches_reference(oracle_ot, case):
    g = load_golden(f"ot_solve_{case}.npz")
    cfg = solve_cfg(g)
    C = _cost(oracle_ot, g)
    G = g["G"] if g["G"].size else np.ones(C.shape[0])
    kw = {k: cfg[k] for k in ("lambda1", "lambda2", "epsilon", "batch_size", "tolerance", "tau",
                              "epsilon0", "max_iter")}
    plan, info = oracle_ot.optimal_transport_duality_gap(C, G, return_info=True, **kw)
    # reference fp64 vs restated fp64: same arithmetic, summation order equal up to
    # compiler vectorisation -> 1e-9 relative on every entry that matters
    np.testing.assert_allclose(plan, g["gamma"], rtol=1e-9, atol=1e-300)
    assert info["stage_iters"].tolist() == g["stage_iters"].tolist()


@pytest.mark.parametrize("case", ["train10x10", "growth64x48"])
def test_compute_transport_map_returns_first_growth_solve(oracle_ot, case):
    g = load_golden(f"ot_solve_{case}.npz")
    cfg = solve_cfg(g)
    G = g["G"] if g["G"].size else None
    full = oracle_ot.compute_transport_map(g["a"], g["b"], cfg, G=G, all_growth_iters=True)
    first = oracle_ot.compute_transport_map(g["a"], g["b"], cfg, G=G, all_growth_iters=False)
    np.testing.assert_allclose(full, g["gamma"], rtol=1e-9)
    np.testing.assert_array_equal(full, first)


def test_entry_points_match_reference(oracle_ot):
    g = load_golden("ot_entry_points.npz")
    C, u, v, eps = g["C"], g["u"], g["v"], float(g["eps"])
    K = np.zeros_like(C); Kb = np.zeros_like(C)
    oracle_ot.update_K(K, Kb, C, u, v, eps)
    np.testing.assert_allclose(K, g["K"], rtol=1e-14)
    np.testing.assert_allclose(Kb, g["Kbar"], rtol=1e-14)
    R = np.zeros_like(C)
    oracle_ot.update_R(R, K, g["a"], g["b"])
    np.testing.assert_allclose(R, g["R"], rtol=1e-14)
    args = (C, g["Kbar"], g["R"], g["dx"], g["dy"], g["p"], g["q"], g["a"], g["b"], eps,
            float(g["l1"]), float(g["l2"]))
    assert oracle_ot.primal(*args) == pytest.approx(float(g["primal"]), rel=1e-13)
    assert oracle_ot.dual(*args) == pytest.approx(float(g["dual"]), rel=1e-12)
    assert oracle_ot.duality_gap(*args) == pytest.approx(float(g["gap"]), rel=1e-12)
    argz = (C, g["Kbar"], g["Rz"]) + args[3:]
    assert oracle_ot.primal(*argz) == pytest.approx(float(g["primal_z"]), rel=1e-13)


def _run_step1(oracle_ot, g, tau, iters, max_iter=10 ** 7):
    m, n = g["C"].shape
    eps, l1, l2 = float(g["eps"]), float(g["l1"]), float(g["l2"])
    a, b, oa, ob = np.ones(m), np.ones(n), np.ones(m), np.ones(n)
    K, u, v = g["K"].copy(), g["u"].copy(), g["v"].copy()
    ret = oracle_ot.step1_process(a, b, oa, ob, K, g["C"], g["dx"], g["dy"], g["p"], g["q"], u, v,
                                  0, max_iter, iters, tau, l1, l2, l1 / (l1 + eps), l2 / (l2 + eps), eps)
    return dict(a=a, b=b, old_a=oa, old_b=ob, K=K, u=u, v=v, ret=ret)


@pytest.mark.parametrize("tag,tau", [("s_no", 1000.0), ("s_ab", 1.05)])
def test_step1_matches_reference(oracle_ot, tag, tau):
    g = load_golden("ot_entry_points.npz")
    out = _run_step1(oracle_ot, g, tau, 5)
    assert out["ret"] == int(g[f"{tag}_ret"])
    for k in ("a", "b", "old_a", "old_b", "K", "u", "v"):
        np.testing.assert_allclose(out[k], g[f"{tag}_{k}"], rtol=1e-12, atol=1e-300, err_msg=k)
    if tag == "s_ab":
        assert not np.array_equal(out["u"], g["u"])  # the fixture did absorb


def test_step1_max_iter_returns_minus_one(oracle_ot):
    g = load_golden("ot_entry_points.npz")
    out = _run_step1(oracle_ot, g, 1000.0, 5, max_iter=1)
    assert out["ret"] == int(g["ret_max"]) == -1


@pytest.mark.parametrize("tag,cur,thr", [("pm", 2, 1e-6), ("pl", 5, 1e-8)])
def test_update_process_matches_reference(oracle_ot, tag, cur, thr):
    g = load_golden("ot_entry_points.npz")
    m, n = g["C"].shape
    eps, l1, l2 = float(g["eps"]), float(g["l1"]), float(g["l2"])
    a, b, oa, ob = np.ones(m), np.ones(n), np.ones(m), np.ones(n)
    K, u, v, R = g["K"].copy(), g["u"].copy(), g["v"].copy(), np.zeros((m, n))
    gap, iters = oracle_ot.update_process(R, a, b, oa, ob, K, g["Kbar"], g["C"], g["dx"], g["dy"],
                                          g["p"], g["q"], u, v, 5, cur, 5, eps, thr, 1000.0, l1, l2,
                                          l1 / (l1 + eps), l2 / (l2 + eps), 0, 10 ** 7)
    assert iters % 5 == 0 and iters > 0
    for k, x in dict(a=a, b=b, old_a=oa, old_b=ob, K=K, u=u, v=v, R=R).items():
        np.testing.assert_allclose(x, g[f"{tag}_{k}"], rtol=1e-10, atol=1e-300, err_msg=k)
    # the converged measure is a difference of nearly equal numbers: compare its size only
    assert gap <= thr and float(g[f"{tag}_gap"]) <= thr


def test_against_compiled_reference_library(oracle_ot):
    """Build-container check: oracle vs the reference's own C++ (oracle/_ref), random inputs."""
    L = oracle_ot.ref_lib()
    if L is None:
        pytest.skip("oracle/_ref/libot_ref.so not built")
    rng = np.random.default_rng(7)
    D = ctypes.POINTER(ctypes.c_double)
    P = lambda x: x.ctypes.data_as(D)
    cd = ctypes.c_double
    for (m, n) in [(5, 9), (64, 33), (130, 257)]:
        C = rng.uniform(0, 4, size=(m, n)); u = rng.normal(size=m) * .2; v = rng.normal(size=n) * .2
        eps, l1, l2, tau = 0.2, 0.1, 5.0, 3.0
        al1, al2 = l1 / (l1 + eps), l2 / (l2 + eps)
        dx, dy = np.ones(m) / m, np.ones(n) / n
        p = rng.uniform(.5, 2, size=m); q = np.ones(n) * p.mean()
        outs = []
        for which in ("ref", "orc"):
            K = np.zeros((m, n)); Kb = np.zeros((m, n)); R = np.zeros((m, n))
            a, b, oa, ob = np.ones(m), np.ones(n), np.ones(m), np.ones(n)
            uu, vv = u.copy(), v.copy()
            if which == "ref":
                L.update_k_double(P(K), P(Kb), P(C), P(uu), P(vv), cd(eps), m, n)
                gap = L.update_process_double(P(R), P(a), P(b), P(oa), P(ob), P(K), P(Kb), P(C), P(dx),
                                              P(dy), P(p), P(q), P(uu), P(vv), 5, 5, 5, cd(eps), cd(1e-8),
                                              cd(tau), cd(l1), cd(l2), cd(al1), cd(al2), 0, 10 ** 7, m, n)
            else:
                oracle_ot.update_K(K, Kb, C, uu, vv, eps)
                gap, _ = oracle_ot.update_process(R, a, b, oa, ob, K, Kb, C, dx, dy, p, q, uu, vv, 5, 5, 5,
                                                  eps, 1e-8, tau, l1, l2, al1, al2, 0, 10 ** 7)
            outs.append((K, R, a, b, uu, vv, gap))
        for x, y in zip(outs[0][:-1], outs[1][:-1]):
            np.testing.assert_allclose(y, x, rtol=1e-10, atol=1e-300)
        assert not np.array_equal(outs[0][4], u)  # tau=3 forces an absorb on this input
