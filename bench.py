#!/usr/bin/env python3
"""bench.py -- hot-path benchmark on MI355X (contract: round prompt section 4).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[2] (SURVEY 8d "cfg3"): 50k spots x 3k genes x 5 time points, i.e.
N_t = 10k spots per time point, G = 3000, 1200 inducing points, kNN k = 30, batches of 512 seeds,
bf16 compute in the GAT branch / fp32 parameters / fp64 SVGP algebra, synthetic data (seed 1993).

The metric has two halves and both are measured, inputs resident in HBM:
  * training steps/s  -- `value`: one step = one NeighborLoader-style batch: forward + backward +
    clip_grad_norm + AdamW (_train_utils.py:187-217), all loss terms active (epoch >= ot_epoch);
  * Sinkhorn iters/s  -- `sinkhorn.value`: one iteration = one update_a_b (ot_func.cpp:586-687) of the
    N_t x N_t coupling between consecutive time points (fp32 kernel matrix, fp64 scalings).
Multi-GPU ("weak"): rank r owns time point r mod 5 (its data, graph, SVGP constants) and the pair
problem (r mod 4, r mod 4 + 1); every step all-reduces the flat gradient buffer over RCCL (two buckets, the first
beside the end of the backward pass); pair
solves need no collective.  `value` = steps of all ranks / max-over-ranks wall time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "training steps/sec + Sinkhorn iters/sec, 50k spots x 3k genes x 5 timepoints"
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
ITERS_PER_STEP = 10
OT_CFG = dict(lambda1=0.1, lambda2=5.0, epsilon=0.05, epsilon0=1.0, tolerance=1e-8, tau=1000.0,
              batch_size=5, max_iter=10 ** 7, growth_iters=3)   # config.yaml ot_config


def synthetic_latents(n, seed, dim=20, k=10, sigma=0.3):
    rng = np.random.default_rng(seed)
    centres = np.random.default_rng(1993).normal(size=(k, dim))
    return centres[rng.integers(0, k, size=n)] + sigma * rng.normal(size=(n, dim))


# ------------------------------------------------------------------------------ CPU baselines

def cpu_sinkhorn(I, J, budget_s=10.0):
    from oracle import ot_oracle
    rng = np.random.default_rng(0)
    K = rng.uniform(0.01, 1.0, size=(I, J))
    a, b = np.ones(I), np.ones(J)
    dx, dy = np.ones(I) / I, np.ones(J) / J
    p, q, u, v = np.ones(I), np.ones(J), np.zeros(I), np.zeros(J)
    eps, l1, l2 = 0.05, 0.1, 5.0
    args = (a, b, K, dx, dy, p, q, u, v, l1, l2, l1 / (l1 + eps), l2 / (l2 + eps), eps)
    ot_oracle.update_a_b(*args)
    n, t0 = 0, time.perf_counter()
    while True:
        ot_oracle.update_a_b(*args)
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 3) or n >= 200:
            break
    return {"value": n / el, "unit": "Sinkhorn iters/s", "cores": 1, "kind": "port",
            "sample": f"{n} update_a_b iterations of one {I}x{J} fp64 problem (oracle/ot_oracle.c, 1 thread), {el:.1f} s"}


def cpu_train_step(model, dd, cfg, tp, bi, tp_prev):
    """ONE training step of the same batch on the host: oracle/model_oracle.training_step (fp64,
    torch CPU with all cores), the reference's arithmetic including its (b, m, m) ELBO tensor."""
    import torch
    from oracle import model_oracle as mo
    batch = dd["dataloaders"][tp][bi]
    loc, Y, _ = dd["datasets"][tp]
    n_id = batch.n_id
    x = loc[n_id].cpu().double()
    y = Y[n_id].float().cpu().double()
    g = batch.graph
    tgt = torch.repeat_interleave(torch.arange(g.n), (g.rowptr[1:] - g.rowptr[:-1]).cpu().long())
    ei = torch.stack([g.col.cpu().long(), tgt])
    P = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    for k in P:
        if P[k].is_floating_point() and "running" not in k:
            P[k].requires_grad_(True)
    sv = mo.SVGPOracle(dd["inducing_points"][tp], dd["N_train"][tp])
    b = batch.batch_size
    gen = torch.Generator().manual_seed(0)
    n1 = torch.randn((b, 10), dtype=torch.float64, generator=gen)
    n2 = torch.randn((b, 10), dtype=torch.float64, generator=gen)
    seeds = n_id[:b].cpu().numpy()
    labels = np.asarray(model.kmeans_cluster_dict[tp])
    km = (model.kmeans_center_dict[tp], labels[seeds])
    ot = (labels[seeds], labels, model.kmeans_center_dict[tp], model.kmeans_center_dict[tp_prev],
          model.gammas[f"{tp_prev}_{tp}"])
    w = (cfg["lambda1"], 0.5, cfg["beta2"], cfg["omiga1"], cfg["omiga2"], cfg["omiga3"])
    t0 = time.perf_counter()
    mo.training_step(P, sv, x, y, ei, b, cfg["gat_attention_heads"], n1, n2, w, km=km, ot=ot, lr=cfg["lr"])
    el = time.perf_counter() - t0
    return {"value": 1.0 / el, "unit": "training steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 training step (batch of {b} seeds, n_sub={g.n}, E={g.E}, m={sv.z.shape[0]}) in fp64 on torch-CPU, "
                      f"{el:.1f} s; oracle/model_oracle.py"}


# ------------------------------------------------------------------------------ main

def main():
    # stdout carries exactly ONE line (the JSON); everything the library prints goes to stderr
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    try:
        _main(real_stdout)
    finally:
        sys.stdout = real_stdout


def _main(real_stdout):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spots", type=int, default=10000, help="spots per time point (N_t)")
    ap.add_argument("--genes", type=int, default=3000)
    ap.add_argument("--timepoints", type=int, default=5)
    ap.add_argument("--compute-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--ot-storage", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--leg", default="both", choices=["both", "train", "sinkhorn"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal switches (one-GPU box): SPADOT_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # SPADOT_BENCH_BACKEND=gloo carries the collectives; the driver's multi-GPU runs use neither
    if os.environ.get("SPADOT_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("SPADOT_BENCH_BACKEND", "nccl")
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    T, N, G = args.timepoints, args.spots, args.genes
    out = {"metric": METRIC, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    cdt = torch.bfloat16 if args.compute_dtype == "bf16" else torch.float32

    # ========================================================================== training leg
    train_res = None
    if args.leg in ("both", "train"):
        import types
        from spadot_amd.synthetic import make_dataset
        from spadot_amd.utils import _train_utils as tu, _utils
        from spadot_amd.model import SpaDOT
        from spadot_amd.ops import FlatAdamW
        cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
        # N == 1: the whole 5-time-point job on one GPU.  N > 1: rank r owns time point r mod T and its
        # predecessor (needed for the OT term); the model is replicated, gradients are all-reduced.
        if world == 1:
            own = list(range(T))
        else:
            t_own = 1 + (rank % (T - 1))
            own = [t_own - 1, t_own]
        data = make_dataset(T, N, G, seed=1993)
        cfg.update(input_dim=G, timepoints=list(range(T)), device=torch.device(dev), compute_dtype=cdt,
                   owned_timepoints=own)
        _utils.set_seed(cfg["seed"])
        t_setup = time.perf_counter()
        dd = tu.prepare_dataloader(data, cfg)
        del data
        model = SpaDOT.SpaDOT(cfg, dd).to(dev)
        opt = FlatAdamW(model.parameters(), lr=cfg["lr"], last=model.GATEncoder.first_layer_parameters())
        tu._update_Kmeans(model, cfg, dd)          # labels/centres so the K-means and OT terms are live
        tu._update_OT_matrix(model, cfg)
        setup_s = time.perf_counter() - t_setup
        model.train()
        grad_sync = grad_sync_async = None
        if world > 1:       # two buckets, same order on every rank and path (spadot_amd.parallel.make_grad_sync)
            from spadot_amd.parallel import make_grad_sync
            grad_sync, grad_sync_async = make_grad_sync(opt)
        # schedule: (tp_i, batch) round robin over the time points that have a predecessor on this rank
        train_tps = [t for t in own if t >= 1 and (t - 1) in own]
        sched = [(t, bi) for bi in range(len(dd["dataloaders"][train_tps[0]])) for t in train_tps]
        epoch = cfg["ot_epoch"]                     # every loss term active
        beta1 = 0.5

        # Steady-state training replays one captured hipGraph per (time point, batch) (epochs >= 2 of a real
        # run); the keys the timed region touches are visited twice beforehand (eager, then capture).
        # Multi-rank: the same staged graphs; the flat gradient crosses RCCL in two buckets (not captured), the first
        # beside the backward pass of the first GAT layer, then one clip + AdamW graph.
        use_graphs = os.environ.get("SPADOT_BENCH_NO_GRAPHS") != "1"
        stepper = (tu.GraphedStepper(model, opt, cfg, dd, grad_sync=grad_sync, grad_sync_async=grad_sync_async)
                   if use_graphs else None)

        state = {"stepper": stepper}

        def step(i):
            t, bi = sched[i % len(sched)]
            if state["stepper"] is not None:
                try:
                    return state["stepper"].step(t, t, bi, epoch, beta1)
                except RuntimeError as ex:          # a capture the runtime refuses: keep measuring, eagerly
                    print(f"[bench] rank {rank}: hipGraph path disabled ({str(ex)[:200]})", file=sys.stderr)
                    state["stepper"] = None
                    torch.cuda.synchronize()
            return tu.training_step(model, opt, cfg, dd, t, t, bi, epoch, beta1, grad_sync=grad_sync)

        if stepper is not None:
            t_cap = time.perf_counter()
            for rep in range(2):
                for i in range(min(len(sched), args.warmup + args.steps)):
                    step(i)
            torch.cuda.synchronize()
            setup_s += time.perf_counter() - t_cap
        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = step(args.warmup + i)
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        b0 = dd["dataloaders"][train_tps[0]][0]
        train_res = {"value": world * args.steps / el, "ms_per_step": 1e3 * el / args.steps,
                     "setup_s": setup_s, "n_sub": b0.graph.n, "E_sub": b0.graph.E,
                     "m_inducing": int(dd["inducing_points"][train_tps[0]].shape[0]),
                     "params": int(opt.count), "last_losses": [float(v) for v in last.cpu().tolist()],
                     "hip_graphs": state["stepper"] is not None,
                     "staged_graphs": bool(state["stepper"] is not None and state["stepper"].staged),
                     "bucketed_grad_exchange": bool(state["stepper"] is not None and state["stepper"].overlap)}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            t, bi = sched[0]
            train_res["cpu_baseline"] = cpu_train_step(model, dd, cfg, t, bi, t - 1)
        del model, opt, dd
        torch.cuda.empty_cache()

    # ========================================================================== Sinkhorn leg
    sk_res, roof = None, None
    if args.leg in ("both", "sinkhorn"):
        from spadot_amd.ot import OTSolver
        I = J = N
        solver = OTSolver(I, J, storage=args.ot_storage, device=dev)
        solver.set_cost_from_latents(synthetic_latents(I, 100 + rank), synthetic_latents(J, 200 + rank))
        t0 = time.perf_counter()
        info = solver.solve(OT_CFG)                 # whole 6-stage solve; leaves a converged state
        torch.cuda.synchronize()
        solve_s = time.perf_counter() - t0
        for _ in range(args.warmup):
            solver.run_iterations(OT_CFG, OT_CFG["epsilon"], ITERS_PER_STEP, timed=False)
        barrier()
        t0 = time.perf_counter()
        ev_ms = 0.0
        for _ in range(args.steps):
            ev_ms += solver.run_iterations(OT_CFG, OT_CFG["epsilon"], ITERS_PER_STEP, timed=True)
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        iters = args.steps * ITERS_PER_STEP
        # the same iterations inside the solver's real loop: + snapshot, duality-gap measure and one
        # host sync every batch_size (5) iterations
        ck_it, ck_ms = solver.run_checked(OT_CFG, OT_CFG["epsilon"], nbatches=max(4, args.steps), last_stage=True)
        esize = 4 if args.ot_storage == "f32" else 8
        kt = solver.time_kernels(OT_CFG, OT_CFG["epsilon"], reps=20)
        geo = solver.fused_geometry()
        # algorithmic bytes per launch = ONE sweep of the I x ld kernel matrix ("roofline");
        # the fused pass's fp64 column partials (workgroups x ld x 8 B) are not counted
        alg = float(I) * solver.ld * esize
        dom = "fused_pass" if geo["vpt"] > 0 else max(("row_pass", "col_pass"), key=lambda k: kt[k])
        ach = alg / (kt[dom] * 1e-3) / 1e9
        sk_res = {"value": world * iters / el, "unit": "Sinkhorn iters/s", "ms_per_iter": 1e3 * el / iters,
                  "event_ms_per_iter": ev_ms / iters, "problem": f"{I}x{J}", "storage": args.ot_storage,
                  "full_solve_s": solve_s, "full_solve_iters": int(sum(info.stage_iters)),
                  "iters_per_s_with_convergence_checks": ck_it / (ck_ms * 1e-3)}
        roof = {"bound": "hbm", "kernel": "k_" + dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": alg,
                "kernel_ms": kt, "fused_geometry": geo}
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2 for 16-byte/lane
        # streams on gfx950, + WRITE_SIZE; MI355X_MICROARCH.md "HBM"): taken from the committed summary of the
        # same workload, when there is one
        pmc = os.path.join(ROOT, "profiles", "r01", "sinkhorn_cfg3_f32_pmc_summary.csv")
        if dom == "fused_pass" and I == 10000 and args.ot_storage == "f32" and os.path.exists(pmc):
            import csv
            for row in csv.reader(open(pmc)):
                if row and "k_fused_pass<float, 5, 2" in row[0]:
                    roof["traffic"] = (2.0 * float(row[1]) + float(row[3])) * 1024.0
                    roof["traffic_source"] = "profiles/r01/sinkhorn_cfg3_f32_pmc_summary.csv (2*FETCH_SIZE + WRITE_SIZE, bytes per launch)"
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            sk_res["cpu_baseline"] = cpu_sinkhorn(I, J)
        solver.close()

    if rank == 0:
        if train_res is not None:
            out.update(value=train_res["value"], unit="training steps/s", ms_per_step=train_res["ms_per_step"])
        else:
            out.update(value=sk_res["value"], unit="Sinkhorn iters/s", ms_per_step=sk_res["ms_per_iter"] * ITERS_PER_STEP)
        out["dtype"] = (f"{args.compute_dtype} GAT branch + linears, f32 parameters/optimizer, f64 SVGP algebra; "
                        f"Sinkhorn {args.ot_storage} kernel matrix with f64 scalings")
        name = "cfg3" if (T, N, G) == (5, 10000, 3000) else "custom shape"
        out["config"] = {"workload": f"{name}: {T} time points x {N} spots x {G} genes, batch 512, k=30, 1200 inducing points; "
                                     f"Sinkhorn pair problem {N}x{N}",
                         "train": {k: v for k, v in (train_res or {}).items() if k != "cpu_baseline"},
                         "parallelism": "1 GPU" if world == 1 else f"{world} ranks: time points / pair problems sharded, "
                                                                   "flat-gradient all-reduce (RCCL) in two buckets, the first overlapped with the backward pass"}
        if sk_res is not None:
            out["sinkhorn"] = {k: v for k, v in sk_res.items() if k != "cpu_baseline"}
            out["roofline"] = roof
        cb = (train_res or {}).get("cpu_baseline") or (sk_res or {}).get("cpu_baseline")
        if cb:
            out["cpu_baseline"] = cb
            if train_res and sk_res and "cpu_baseline" in sk_res and "cpu_baseline" in train_res:
                out["cpu_baseline_sinkhorn"] = sk_res["cpu_baseline"]
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
