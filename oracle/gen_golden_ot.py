"""oracle/gen_golden_ot.py -- generates tests/golden/ot_*.npz FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  It imports the reference's
``ot_func.py`` / ``ot_solvers.py`` by path (parent packages are registered as empty
namespace modules so that ``SpaDOT/__init__.py`` -- which needs scanpy/anndata -- is not
executed; POT's ``ot`` module, used only by a function nobody calls, is registered empty)
and records inputs and outputs of the reference's own functions.  The fixtures are data
(inputs + expected outputs); no reference source is written anywhere.

    python oracle/gen_golden_ot.py
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np
import yaml

REF = "/root/reference/SpaDOT"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference():
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    ns("SpaDOT", REF)
    ns("SpaDOT.utils", REF + "/utils")
    ns("SpaDOT.utils.OT_loss", REF + "/utils/OT_loss")
    sys.modules.setdefault("ot", types.ModuleType("ot"))

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    f = load("SpaDOT.utils.OT_loss.ot_func", REF + "/utils/OT_loss/ot_func.py")
    s = load("SpaDOT.utils.OT_loss.ot_solvers", REF + "/utils/OT_loss/ot_solvers.py")
    return f, s


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def mixture(rng, n, d=20, k=10, sigma=0.3, centres=None):
    if centres is None:
        centres = rng.normal(size=(k, d))
    lab = rng.integers(0, centres.shape[0], size=n)
    return centres[lab] + sigma * rng.normal(size=(n, d)), centres


def main():
    ot_func, ot_solvers = load_reference()
    base_cfg = yaml.safe_load(open(REF + "/config.yaml"))["ot_config"]
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(1993)

    # ---------------- (1) whole solves through compute_transport_map -----------------
    def solve_case(name, a, b, cfg, G=None, C=None):
        cfg = dict(cfg)
        # the call under test: exactly what _train_utils.py:318 does
        gamma = quiet(ot_solvers.compute_transport_map, a, b, dict(cfg), C=C, G=G)
        Cm = cfg_C = None
        # what the reference built as cost (it stores it in the dict it is given)
        probe = dict(cfg)
        quiet(ot_solvers.compute_transport_map, a, b, probe, C=C, G=G)
        Cm = np.array(probe["C"])
        # iteration counts per stage: re-run the first solve on the reference's own
        # python while-loop around its C step1/gap functions (ot_solvers.py:291-433)
        # and count step1 iterations; identical arithmetic to the c_for_v2 path.
        counts, absorbed = [], []
        orig = ot_solvers.step1_process_c

        def counting(a_, b_, oa, ob, K, C_, dx, dy, p, q, u, v, cur, mx, iters, *rest):
            u0, v0 = u.copy(), v.copy()
            r = orig(a_, b_, oa, ob, K, C_, dx, dy, p, q, u, v, cur, mx, iters, *rest)
            counts.append(iters)
            # absorb rewrites u, v in place (ot_func.cpp:792-800)
            absorbed.append(not (np.array_equal(u0, u) and np.array_equal(v0, v)))
            return r

        stage_marks = []
        orig_k = ot_solvers.update_K_c

        def marking(*args, **kw):
            stage_marks.append(len(counts))
            return orig_k(*args, **kw)

        ot_solvers.step1_process_c = counting
        ot_solvers.update_K_c = marking
        ot_solvers.c_for_v2 = False
        try:
            g0 = np.ones(Cm.shape[0]) if G is None else np.asarray(G, dtype=np.float64)
            kw = dict(cfg)
            kw.update(C=Cm, G=g0)
            gamma_loop = quiet(ot_solvers.optimal_transport_duality_gap, **kw)
        finally:
            ot_solvers.step1_process_c = orig
            ot_solvers.update_K_c = orig_k
            ot_solvers.c_for_v2 = True
        stage_marks.append(len(counts))
        stage_iters = [int(sum(counts[stage_marks[i]:stage_marks[i + 1]])) for i in range(6)]
        assert np.allclose(gamma_loop, gamma, rtol=1e-12, atol=0), name
        keys = ("lambda1", "lambda2", "epsilon", "batch_size", "tolerance", "tau", "epsilon0",
                "max_iter", "growth_iters")
        np.savez_compressed(
            os.path.join(OUT, f"ot_solve_{name}.npz"),
            a=np.asarray(a) if a is not None else np.zeros(0),
            b=np.asarray(b) if b is not None else np.zeros(0),
            C=Cm if Cm.size <= 20000 or a is None else np.zeros(0), G=np.zeros(0) if G is None else np.asarray(G, dtype=np.float64),
            gamma=gamma, stage_iters=np.array(stage_iters, dtype=np.int32),
            any_absorb=np.array(any(absorbed)), n_absorb=np.array(sum(absorbed)),
            cfg_keys=np.array(keys), cfg_vals=np.array([float(cfg[k]) for k in keys]))
        print(f"ot_solve_{name}: {Cm.shape} iters/stage {stage_iters} absorb={sum(absorbed)} "
              f"sum={gamma.sum():.6g}")

    # training shape: 10 k-means centres per time point (_train_utils.py:318-320)
    c1 = rng.normal(size=(10, 20)); c2 = c1 + 0.4 * rng.normal(size=(10, 20))
    solve_case("train10x10", c1, c2, base_cfg)
    # ragged
    solve_case("ragged7x13", rng.normal(size=(7, 20)), rng.normal(size=(13, 20)), base_cfg)
    # single row / single column edge cases
    solve_case("edge1x5", rng.normal(size=(1, 20)), rng.normal(size=(5, 20)), base_cfg)
    # non-uniform growth
    xa, cen = mixture(rng, 64); xb, _ = mixture(rng, 48, centres=cen + 0.1)
    solve_case("growth64x48", xa, xb, base_cfg, G=rng.uniform(0.5, 2.0, size=64))
    # spot-level shape (small): mixture latents
    xa, cen = mixture(rng, 300); xb, _ = mixture(rng, 400, centres=cen + 0.2 * rng.normal(size=cen.shape))
    solve_case("spots300x400", xa, xb, base_cfg)
    # forced stabilisation: tau small so that absorb fires repeatedly
    cfg_tau = dict(base_cfg); cfg_tau["tau"] = 1.5
    xa, cen = mixture(rng, 120); xb, _ = mixture(rng, 150, centres=cen + 0.3)
    solve_case("absorb120x150", xa, xb, cfg_tau)
    # default tau with an outlier row and the solver module's own defaults (lambda1=1, lambda2=50)
    cfg_out = dict(base_cfg); cfg_out["lambda1"] = 1.0; cfg_out["lambda2"] = 50.0
    xa, cen = mixture(rng, 40); xb, _ = mixture(rng, 56, centres=cen)
    xa[3] += 9.0
    solve_case("outlier40x56", xa, xb, cfg_out)

    # ---------------- (2) individual C entry points on random positive inputs ---------
    m, n = 23, 37
    C = rng.uniform(0.0, 3.0, size=(m, n))
    u = 0.1 * rng.normal(size=m); v = 0.1 * rng.normal(size=n)
    eps = 0.3
    K = np.zeros((m, n)); Kb = np.zeros((m, n))
    ot_func.update_K_c(K, Kb, C, u, v, eps)
    a = rng.uniform(0.5, 2.0, size=m); b = rng.uniform(0.5, 2.0, size=n)
    R = np.zeros((m, n))
    ot_func.update_R_c(R, K, a, b)
    dx = np.ones(m) / m; dy = np.ones(n) / n
    p = rng.uniform(0.5, 2.0, size=m); q = np.ones(n) * p.mean()
    l1, l2 = 0.1, 5.0
    pri = ot_func.primal_c(C, Kb, R, dx, dy, p, q, a, b, eps, l1, l2)
    dua = ot_func.dual_c(C, Kb, R, dx, dy, p, q, a, b, eps, l1, l2)
    gap = ot_func.compute_duality_gap_c(C, Kb, R, dx, dy, p, q, a, b, eps, l1, l2)
    # a plan with exact zeros exercises the log(0) clamp (ot_func.cpp:29-40, :414)
    Rz = R.copy(); Rz[::3, ::4] = 0.0
    pri_z = ot_func.primal_c(C, Kb, Rz, dx, dy, p, q, a, b, eps, l1, l2)

    # step1: no absorb (tau large) and absorb (tau small)
    def run_step1(tau, iters):
        a1, b1 = np.ones(m), np.ones(n)
        oa, ob = np.ones(m), np.ones(n)
        K1 = K.copy(); u1 = u.copy(); v1 = v.copy()
        al1 = l1 / (l1 + eps); al2 = l2 / (l2 + eps)
        ret = ot_func.step1_process_c(a1, b1, oa, ob, K1, C, dx, dy, p, q, u1, v1, 0, 10 ** 7,
                                      iters, tau, l1, l2, al1, al2, eps)
        return dict(a=a1, b=b1, old_a=oa, old_b=ob, K=K1, u=u1, v=v1, ret=np.array(ret))

    s_no = run_step1(1000.0, 5)
    s_ab = run_step1(1.05, 5)
    s_max = None
    # max_iter reached: returns -1 after the first iteration (ot_func.cpp:821-824)
    a1, b1, oa, ob = np.ones(m), np.ones(n), np.ones(m), np.ones(n)
    K1 = K.copy(); u1 = u.copy(); v1 = v.copy()
    ret_max = quiet(ot_func.step1_process_c, a1, b1, oa, ob, K1, C, dx, dy, p, q, u1, v1, 0, 1, 5,
                    1000.0, l1, l2, l1 / (l1 + eps), l2 / (l2 + eps), eps)

    # update_process: a middle stage (drift criterion) and the last stage (duality gap)
    def run_process(cur_scaling, thr, batch):
        a1, b1 = np.ones(m), np.ones(n)
        oa, ob = np.ones(m), np.ones(n)
        K1 = K.copy(); u1 = u.copy(); v1 = v.copy(); R1 = np.zeros((m, n))
        al1 = l1 / (l1 + eps); al2 = l2 / (l2 + eps)
        g = ot_func.update_process_c(R1, a1, b1, oa, ob, K1, Kb, C, dx, dy, p, q, u1, v1, 5,
                                     cur_scaling, batch, eps, thr, 1000.0, l1, l2, al1, al2, 0, 10 ** 7)
        return dict(a=a1, b=b1, old_a=oa, old_b=ob, K=K1, u=u1, v=v1, R=R1, gap=np.array(g))

    pm = run_process(2, 1e-6, 5)
    pl = run_process(5, 1e-8, 5)

    np.savez_compressed(
        os.path.join(OUT, "ot_entry_points.npz"),
        C=C, u=u, v=v, eps=np.array(eps), K=K, Kbar=Kb, a=a, b=b, R=R, dx=dx, dy=dy, p=p, q=q,
        l1=np.array(l1), l2=np.array(l2), primal=np.array(pri), dual=np.array(dua),
        gap=np.array(gap), Rz=Rz, primal_z=np.array(pri_z), ret_max=np.array(ret_max),
        **{f"s_no_{k}": x for k, x in s_no.items()},
        **{f"s_ab_{k}": x for k, x in s_ab.items()},
        **{f"pm_{k}": x for k, x in pm.items()},
        **{f"pl_{k}": x for k, x in pl.items()})
    print("ot_entry_points: primal", pri, "dual", dua, "gap", gap, "step1 absorb ret", s_ab["ret"],
          "ret_max", ret_max, "pm gap", pm["gap"], "pl gap", pl["gap"])

    # ---------------- (3) added in round 2, own RNG so that the fixtures above keep their bytes -------------
    # last-stage batches LONGER than the device solver's 64-iteration chunk, with a tau small enough that the
    # stabilisation fires inside such a batch (regression: a per-chunk flag clear used to lose it)
    rng2 = np.random.default_rng(2024)
    # (lambda 1 / 50: the scalings keep growing through the last stage; with tau = 2.4 the stabilisation fires at
    # iteration 85 of the first 200-iteration batch of the last stage, i.e. in its second 64-iteration chunk, and not in
    # the chunks after it -- traced with oracle/ot_oracle.py step by step; stages 0 and 4 stabilise too)
    cfg_long = dict(base_cfg); cfg_long.update(tau=2.4, batch_size=200, lambda1=1.0, lambda2=50.0)
    xa, cen = mixture(rng2, 90); xb, _ = mixture(rng2, 110, centres=cen + 0.3)
    solve_case("longbatch90x110", xa, xb, cfg_long)


if __name__ == "__main__":
    main()
