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


def write_transition_tables(output_dir, latents, labels, timepoints, prefix="", config=None, which="last", storage="f32",
                            device="cuda:0", write_tmaps=False):
    """The analyze stage's file outputs for the OT part (_analyze_utils.py:124-138) without anndata / wot:
      {prefix}transition_table_{day}_{day+1}.csv / .npz   the aggregated OT matrix between the K-means domains of two
                                                          consecutive time points (rows '<tp>_<cluster>' of the earlier one);
                                                          days are the category codes of the sorted time points (:119)
      OT/tmap_{day}_{day+1}.npz   (write_tmaps=True)      the spot-level transport map itself (fp32, N_t x N_{t+1}) -- what
                                                          wot's compute_all_transport_maps(tmap_out=...) leaves under OT/
      OT_g.txt                                            the growth vector fed to the last solve, one row per spot
    tools/npz_to_h5ad.py turns the .npz tables into the .h5ad files the reference writes (in an environment with anndata).
    Returns the list of tables."""
    import os
    os.makedirs(output_dir, exist_ok=True)
    days = list(range(len(timepoints)))
    tabs = []
    growth_rows = []
    for t in range(len(latents) - 1):
        solver, infos = spot_transport(latents[t], latents[t + 1], config, which=which, storage=storage, device=device)
        try:
            la, lb = np.asarray(labels[t]), np.asarray(labels[t + 1])
            ka, kb = int(la.max()) + 1, int(lb.max()) + 1
            tab = solver.transition_table(la, lb, ka, kb).cpu().numpy()
            rows = np.array([f"{timepoints[t]}_{c}" for c in range(ka)])
            cols = np.array([f"{timepoints[t + 1]}_{c}" for c in range(kb)])
            stem = os.path.join(output_dir, f"{prefix}transition_table_{days[t]}_{days[t + 1]}")
            np.savez_compressed(stem + ".npz", X=tab, obs_names=rows, var_names=cols)
            with open(stem + ".csv", "w") as fh:
                fh.write("," + ",".join(cols.tolist()) + "\n")
                for r, name in enumerate(rows.tolist()):
                    fh.write(name + "," + ",".join(repr(float(v)) for v in tab[r]) + "\n")
            growth_rows.append(solver.plan_rowsums() if which == "last" else np.ones(la.size))
            if write_tmaps:
                os.makedirs(os.path.join(output_dir, "OT"), exist_ok=True)
                np.savez_compressed(os.path.join(output_dir, "OT", f"tmap_{days[t]}_{days[t + 1]}.npz"),
                                    X=solver.plan("torch").float().cpu().numpy())
            tabs.append(tab)
        finally:
            solver.close()
    if growth_rows:
        np.savetxt(os.path.join(output_dir, "OT_g.txt"), np.concatenate(growth_rows))
    return tabs
