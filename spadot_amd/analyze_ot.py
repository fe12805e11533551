"""Spot-level optimal transport of the analyze stage: the N_t x N_{t+1} coupling between the
latent embeddings of consecutive time points and its cluster transition table -- the one place the product
runs Sinkhorn at spot scale (/root/reference/SpaDOT/utils/_analyze_utils.py:108-138, there through the
third-party `wot.ot.OTModel(epsilon=0.05, epsilon0=1, lambda1=0.1, lambda2=5, growth_iters=3)`).

wot is un-vendored and unpinned, so parity is against the solver vendored in the reference
(ot_solvers.py), whose arithmetic this package reproduces: cost = sqeuclidean / median, `growth_iters`
solves with the previous plan's row sums fed back as growth.  wot keeps the LAST growth iteration's map
(the vendored compute_transport_map returns the first: SURVEY App. D.1); `which` selects either.
Everything stays in HBM: the plan is never materialised unless asked for, the transition table is a
device reduction (OTSolver.transition_table).
"""
import numpy as np

from .ot import OTSolver

ANALYZE_OT_CONFIG = dict(epsilon=0.05, epsilon0=1.0, lambda1=0.1, lambda2=5.0, tau=1000.0, tolerance=1e-8,
                         batch_size=5, max_iter=10 ** 7, growth_iters=3)   # _analyze_utils.py:124 + wot defaults


def spot_transport(latent_a, latent_b, config=None, growth=None, which="last", storage="f32", device="cuda:0"):
    """Solve the spot-level coupling.  Returns an OTSolver holding the converged plan (call .plan(),
    .plan_rowsums() or .transition_table() on it; close() it when done) and the list of per-solve infos."""
    cfg = dict(ANALYZE_OT_CONFIG if config is None else config)
    I, J = int(np.shape(latent_a)[0]), int(np.shape(latent_b)[0])
    solver = OTSolver(I, J, storage=storage, device=device)
    solver.set_cost_from_latents(latent_a, latent_b, divide_by_median=True)
    g = np.ones(I) if growth is None else np.asarray(growth, dtype=np.float64)
    infos = []
    n = int(cfg.get("growth_iters", 1)) if which == "last" else 1
    for i in range(n):
        if i > 0:
            g = solver.plan_rowsums()
        infos.append(solver.solve(cfg, g))
    return solver, infos


def transition_tables(latents, labels, config=None, which="last", storage="f32", device="cuda:0"):
    """latents / labels: lists over time points (arrays [N_t, z_dim] / integer cluster ids [N_t]).
    Returns [(table [K_t, K_{t+1}] numpy fp64, infos), ...] for consecutive pairs."""
    out = []
    for t in range(len(latents) - 1):
        solver, infos = spot_transport(latents[t], latents[t + 1], config, which=which, storage=storage, device=device)
        try:
            tab = solver.transition_table(labels[t], labels[t + 1])
            out.append((tab.cpu().numpy(), infos))
        finally:
            solver.close()
    return out
