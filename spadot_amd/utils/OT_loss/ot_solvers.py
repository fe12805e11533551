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

from ...ot import CONFIG_KEYS, OTSolver, small_problem_ok, solve_small

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
# problems with I, J <= 64 (the training loop's 10 x 10 couplings of K-means centres, _train_utils.py:309-321) take the
# single-launch solver of csrc/ot_small.hip: same arithmetic and stopping rules, no host synchronisation inside the solve
use_small_solver = True

MAX_ITER_MESSAGE = "Reached max_iter with duality gap still above threshold. Returning"   # ot_func.cpp:821-824


def _small_results(res):
    """Bookkeeping shared by the small-solver paths: last_info, the reference's max_iter print, its NaN error.
    Returns False when a problem hit the kernel's iteration cap (the caller then uses the streaming solver)."""
    global last_info
    for info in res.infos:
        if info.status & 2:
            return False
    for info in res.infos:
        last_info = info
        if info.status & 1:
            print(MAX_ITER_MESSAGE, end="")
        if np.isnan(info.gap):
            raise RuntimeError("Overflow encountered in duality gap computation, please report this incident")
    return True


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
    if use_small_solver and small_problem_ok(I, J):
        res = solve_small(cfg, costs=[C], growth=[None if G is None else np.asarray(G, dtype=np.float64)],
                          divide_by_median=False, device=device)
        if _small_results(res):
            return res.plans[0]
    solver = OTSolver(I, J, storage=storage, device=device)
    try:
        solver.set_cost(C)
        _solve(solver, cfg, None if G is None else np.asarray(G, dtype=np.float64))
        return solver.plan("numpy")
    finally:
        solver.close()


def compute_transport_map(a, b, config, C=None, G=None, *, device=None, skip_small=False):
    """Transport map between two latent point clouds (ot_solvers.py:95-121).  `device` (extension): the HIP device to
    solve on; default = this module's `device` switch.  `skip_small` (extension): go straight to the streaming solver --
    the fallback of a batched small solve that hit its iteration cap passes it, so that the capped kernel (which cannot
    be stopped: up to 6 x 2^20 iterations on one wavefront) is not spun a second time on the same problem."""
    device = device if device is not None else globals()["device"]
    if C is None:
        xa = torch.as_tensor(a.detach() if isinstance(a, torch.Tensor) else np.asarray(a))
        xb = torch.as_tensor(b.detach() if isinstance(b, torch.Tensor) else np.asarray(b))
        I, J = int(xa.shape[0]), int(xb.shape[0])
    else:
        I, J = C.shape
    if use_small_solver and not skip_small and small_problem_ok(I, J, None if C is not None else int(xa.shape[1])):
        first = _compute_transport_map_small(xa if C is None else None, xb if C is None else None, config, C, G, I, device)
        if first is not None:
            return first
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


def _compute_transport_map_small(xa, xb, config, C, G, I, device):
    """compute_transport_map for I, J <= 64 through the single-launch solver; None if it has to be redone by the
    streaming solver (iteration cap)."""
    config["C"] = C
    config["G"] = np.ones(I) if G is None else G
    cfg = {k: config[k] for k in CONFIG_KEYS}
    first = None
    row_sums = np.asarray(config["G"], dtype=np.float64)
    for i in range(config["growth_iters"]):
        print("OT iter", i)
        config["G"] = row_sums
        if C is None:
            res = solve_small(cfg, pairs=[(xa, xb)], growth=[row_sums], divide_by_median=True, device=device)
        else:
            res = solve_small(cfg, costs=[C], growth=[row_sums], divide_by_median=False, device=device)
        if not _small_results(res):
            return None
        if first is None:
            first = res.plans[0]
            if not run_discarded_growth_iters:
                break
        row_sums = res.plans[0].sum(axis=1)
    return first


def compute_transport_maps(pairs, config, gamma_out=None, device=None):
    """All consecutive-pair couplings of an epoch in ONE launch: `pairs` = [(centres_t, centres_t+1), ...], every one
    within the small solver's range.  Same result per pair as compute_transport_map(a, b, config) (first growth solve,
    G = ones); gamma_out: optional fp32 device tensors that receive the row-normalised plans in place.  Returns the
    list of plans, or None when a problem is out of range / hit the iteration cap (use compute_transport_map then)."""
    if not use_small_solver or not pairs:
        return None
    device = device if device is not None else globals()["device"]
    for a, b in pairs:
        if a.shape[1] != b.shape[1] or not small_problem_ok(a.shape[0], b.shape[0], a.shape[1]):
            return None
    for _ in pairs:
        print("OT iter", 0)
    cfg = {k: config[k] for k in CONFIG_KEYS}
    res = solve_small(cfg, pairs=pairs, divide_by_median=True, gamma_out=gamma_out, device=device)
    if not _small_results(res):
        return None
    config["C"] = None
    config["G"] = np.ones(pairs[-1][0].shape[0])
    return res.plans
