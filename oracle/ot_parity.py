"""oracle/ot_parity.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A whole `optimal_transport_duality_gap` (ot_solvers.py:164-449 -> ot_func.cpp:586-930) of the C oracle on a problem given
by its latents, and the comparison of a device plan with it -- used at BASELINE.json's full size (10 000 x 10 000, the
cfg3 pair problem: about a minute of one host thread) by tests/test_ot_gpu.py and by bench.py's cpu_baseline_sinkhorn
leg.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import this module.

Stated tolerances (reference fp64 vs fp32 storage): per-stage iteration counts equal up to one convergence
check (5 iterations), row / column marginals rtol 1e-4, plan entries > 1e-9 * max rtol 1e-3.
"""
import time

import numpy as np

from . import ot_oracle

SOLVER_KEYS = ("lambda1", "lambda2", "epsilon", "batch_size", "tolerance", "tau", "epsilon0", "max_iter")


def oracle_solve_from_latents(x, y, cfg, G=None):
    """cost = sqeuclidean / median (ot_solvers.py:101-103), then the whole six-stage solve, fp64, one thread.
    Returns (plan R / J, info dict with 'stage_iters' and 'gap', seconds of the solve alone)."""
    C = ot_oracle.sqeuclidean_cost(x, y)
    C /= np.median(C)
    G = np.ones(C.shape[0]) if G is None else np.asarray(G, dtype=np.float64)
    t0 = time.perf_counter()
    plan, info = ot_oracle.optimal_transport_duality_gap(C, G, return_info=True, **{k: cfg[k] for k in SOLVER_KEYS})
    return plan, info, time.perf_counter() - t0


def compare_plans(P, stage_iters, ref, ref_stage_iters):
    """JSON-ready agreement of a device plan with the oracle's."""
    P = np.asarray(P, dtype=np.float64)
    it_d, it_r = [int(v) for v in stage_iters], [int(v) for v in ref_stage_iters]
    rows_d, rows_r = P.sum(axis=1), ref.sum(axis=1)
    cols_d, cols_r = P.sum(axis=0), ref.sum(axis=0)
    big = ref > 1e-9 * ref.max()
    rel = np.abs(P[big] - ref[big]) / ref[big]
    return {
        "stage_iters_dev": it_d, "stage_iters_ref": it_r,
        "stage_iters_equal": it_d == it_r,
        "stage_iters_max_diff": int(max(abs(a - b) for a, b in zip(it_d, it_r))),
        "marginal_rel_err": float(max(np.max(np.abs(rows_d - rows_r) / rows_r), np.max(np.abs(cols_d - cols_r) / cols_r))),
        "plan_rel_err_top": float(rel.max()),
        "plan_rel_err_top_p999": float(np.quantile(rel, 0.999)),
        "entries_compared": int(big.sum()),
        "mass_dev": float(P.sum()), "mass_ref": float(ref.sum()),
    }
