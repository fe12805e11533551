"""Mirror of /root/reference/SpaDOT/utils/OT_loss/ot_solvers.py for the MI355X path.

Same public functions and argument meaning:
  compute_transport_map(a, b, config, C=None, G=None)           ot_solvers.py:95-121
  optimal_transport_duality_gap(C, G, lambda1, ..., **ignored)  ot_solvers.py:164-449
but the six-stage solve runs as ONE call into the device-resident HIP solver (spadot_amd.ot)
instead of 12 numpy<->C crossings per solve, and the cost matrix is built on the GPU.

Behaviour kept from the reference: the config dict passed in gets its "C" and
"G" entries overwritten; `growth_iters` solves are defined with row sums fed back as growth but
only the FIRST plan is returned; a NaN gap raises RuntimeError.  Because solves 2..n never
influence the return value, they are skipped unless ``run_discarded_growth_iters=True``.
"""
import numpy as np
import torch

from ...ot import CONFIG_KEYS, OTSolver

default_config = {
    "growth_iters": 3, "epsilon": 0.05, "lambda1": 1, "lambda2": 50, "epsilon0": 1, "tau": 1000,
    "scaling_iter": 3000, "inner_iter_max": 50, "tolerance": 1e-8, "max_iter": 1e7, "batch_size": 5,
    "extra_iter": 1000, "numItermax": 1000000, "use_Py": False, "use_C": True, "profiling": False,
}  # ot_solvers.py:19-36

# module switches: storage precision of the I x J matrices in HBM and the device to run on
storage = "f64"
device = "cuda:0"
run_discarded_growth_iters = False
last_info = None   # spadot_ot_info of the most recent solve (iterations per stage, absorbs, gap)


def _solve(solver, cfg, G):
    global last_info
    last_info = solver.solve(cfg, G)
    return last_info


def optimal_transport_duality_gap(C, G, lambda1, lambda2, epsilon, batch_size, tolerance, tau, epsilon0,
                                  max_iter, use_Py=False, use_C=True, profiling=False, **ignored):
    """Entropy-regularised unbalanced transport map with duality gap <= tolerance; returns R / J as
    a numpy fp64 array (ot_solvers.py:449).  C may be a numpy array or a (device) torch tensor."""
    if use_Py:
        raise NotImplementedError("use_Py selects the reference's numpy loop; this package only has the HIP path")
    I, J = C.shape
    cfg = dict(lambda1=lambda1, lambda2=lambda2, epsilon=epsilon, batch_size=batch_size, tolerance=tolerance,
               tau=tau, epsilon0=epsilon0, max_iter=max_iter)
    solver = OTSolver(I, J, storage=storage, device=device)
    try:
        solver.set_cost(C)
        _solve(solver, cfg, None if G is None else np.asarray(G, dtype=np.float64))
        return solver.plan("numpy")
    finally:
        solver.close()


def compute_transport_map(a, b, config, C=None, G=None):
    """Transport map between two latent point clouds (ot_solvers.py:95-121)."""
    if C is None:
        xa = torch.as_tensor(a.detach() if isinstance(a, torch.Tensor) else np.asarray(a))
        xb = torch.as_tensor(b.detach() if isinstance(b, torch.Tensor) else np.asarray(b))
        I, J = int(xa.shape[0]), int(xb.shape[0])
    else:
        I, J = C.shape
    solver = OTSolver(I, J, storage=storage, device=device)
    try:
        if C is None:
            solver.set_cost_from_latents(xa, xb, divide_by_median=True)
            config["C"] = None     # the reference stores the ndarray here; ours never leaves HBM
        else:
            solver.set_cost(C)
            config["C"] = C
        config["G"] = np.ones(I) if G is None else G
        cfg = {k: config[k] for k in CONFIG_KEYS}
        growth_iters = config["growth_iters"]
        first = None
        row_sums = config["G"]
        for i in range(growth_iters):
            print("OT iter", i)
            if i > 0:
                row_sums = solver.plan_rowsums()
            config["G"] = row_sums
            _solve(solver, cfg, np.asarray(row_sums, dtype=np.float64))
            if first is None:
                first = solver.plan("numpy")
                if not run_discarded_growth_iters:
                    break
        return first
    finally:
        solver.close()
