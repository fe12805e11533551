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
Multi-GPU ("weak": one batch per rank per step): the job's (time point, batch) units are dealt to the ranks for the
whole run (spadot_amd.parallel, granularity 'batch'), every rank holds all time points' rows; rank r also solves the pair
problem (r mod 4, r mod 4 + 1); every step all-reduces the flat gradient buffer over RCCL (two buckets, the first
beside the end of the backward pass); pair solves need no collective.  `value` = steps of all ranks / max-over-ranks
wall time.

Prints ONE JSON line on rank 0 -- at most LINE_LIMIT (6000) characters: the contract's keys, `roofline`, `cpu_baseline` and
one-number summaries; the full record (texts, per-repeat timings, kernel tables) goes to gpurun_out/bench_detail_<N>gpu.json,
named in the line as `detail_file` (tests/test_bench_line_cpu.py).  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE
starts the N ranks itself (children, `python -m torch.distributed.run`), before anything touches the GPU.
"""
import contextlib
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


def cpu_sinkhorn_parity(x, y, plan_dev, stage_iters_dev):
    """ONE whole optimal_transport_duality_gap of the C oracle (fp64, 1 thread) on the benchmarked pair problem, and the
    device plan of that same problem against it (oracle/ot_parity.py; tests/test_ot_gpu.py holds the asserting version:
    per-stage iteration counts within one convergence check, marginals rtol 1e-4, entries > 1e-9 max rtol 1e-3)."""
    from oracle import ot_parity
    ref, rinfo, secs = ot_parity.oracle_solve_from_latents(x, y, OT_CFG)
    rep = ot_parity.compare_plans(plan_dev, stage_iters_dev, ref, rinfo["stage_iters"])
    rep["oracle_solve_s"] = secs
    rep["oracle_iters_per_s_whole_solve"] = float(sum(rep["stage_iters_ref"])) / secs
    rep["what"] = ("device plan (fp32 storage, fp64 scalings) of the benchmarked problem vs one whole six-stage solve of "
                   "oracle/ot_oracle.c (fp64, 1 thread) on the same latents")
    return rep


def cpu_train_steps(model, opt, dd, cfg, tu, tp, bi, tp_prev, epoch, beta1, n_steps=3):
    """`n_steps` training steps of the same batch on the host: oracle/model_oracle.training_step (fp64, torch CPU with
    all cores), the reference's arithmetic including its (b, m, m) ELBO tensor (SURVEY 8d asks for >= 3 steps).
    The first of them doubles as the PARITY CHECK of the benchmarked device step: the eager HIP step body is run on
    the same weights / batch / graph / noise / K-means state / OT plan, and its seven loss terms, final_latent and
    per-parameter gradients are compared with the oracle's (oracle/step_parity.py)."""
    import torch
    from oracle import step_parity as sp
    batch = dd["dataloaders"][tp][bi]
    noise = sp.make_noise(batch.batch_size, seed=0)
    inp = sp.oracle_inputs(model, dd, cfg, tp, bi, tp_prev)              # snapshot of the current weights
    dl, dz, dg = sp.device_step(model, opt, cfg, dd, tu, 1, tp, bi, epoch, beta1, noise)
    ref = sp.oracle_step(inp, cfg, beta1, noise, n_steps=n_steps)
    rep = sp.compare(dl, dz, dg, ref)
    secs = ref["seconds"]
    med = float(np.median(secs))
    cpu = {"value": 1.0 / med, "unit": "training steps/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{len(secs)} training steps (batch of {inp['b']} seeds, n_sub={inp['n_sub']}, E={inp['E']}, m={inp['m']}) "
                     f"in fp64 on torch-CPU, {', '.join(f'{t:.1f}' for t in secs)} s (median {med:.1f} s); oracle/model_oracle.py"}
    keep = ("max_rel_loss_err", "loss_names", "loss_rel_err", "latent_rel_l2_err", "latent_max_abs_err", "grad_cos_min",
            "grad_cos_min_param", "grad_rel_l2_max", "grad_rel_l2_max_param", "grad_cos_global", "zero_grad_dev_rel_norm_max")
    parity = {k: rep[k] for k in keep}
    parity["what"] = ("eager HIP step body vs the fp64 host oracle on the same weights, batch, noise, K-means state and OT "
                      "plan: 7 loss terms, final_latent, gradient of every parameter (tests/test_step_parity_gpu.py holds "
                      "the asserting version)")
    return cpu, parity


# ------------------------------------------------------------------------------ training-leg roofline, epoch block

MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (spec; the guide measures ~1.25-1.5 PF on random data)
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32 class) peak -- what the fp32 compute dtype is priced on


def source_fingerprint():
    """sha256 (first 16 hex digits) over the sources the measured kernels come from: spadot_amd/**/*.{py,hip,yaml},
    include/*.h and this file.  tools/prof_summary.py stores it in the committed profile summaries; `stale_profile` in the
    bench line says whether the tree that runs is the tree that was profiled (a commit that only adds the profile files
    does not change it, unlike `git rev-parse HEAD`)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "bench.py")]
    for base, exts in ((os.path.join(ROOT, "spadot_amd"), (".py", ".hip", ".yaml")), (os.path.join(ROOT, "include"), (".h",))):
        for d, _, names in os.walk(base):
            if "__pycache__" in d:
                continue
            files += [os.path.join(d, n) for n in names if n.endswith(exts)]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def newest_profile(name):
    """Path of profiles/rNN/<name> of the highest round that has it, or None."""
    base = os.path.join(ROOT, "profiles")
    for d in sorted((x for x in os.listdir(base) if x.startswith("r")), reverse=True) if os.path.isdir(base) else []:
        p = os.path.join(base, d, name)
        if os.path.exists(p):
            return p
    return None


def roofline_train(dd, unit, cfg, G, compute_dtype, n_params, live_ms_per_step, tag="cfg3"):
    """Where the training step stands against the machine, per kernel family.

    ALGORITHMIC work per step comes from the shapes of the batch actually run (computed here, live); the MICROSECONDS
    per family cannot be taken from inside a graph replay with events, so they are REPLAYED from the committed
    rocprofv3 --kernel-trace summary of this same command (tools/prof_summary.py -> profiles/rNN/train_cfg3_<dtype>_families.json),
    and the block says so.  Pricing (DESIGN.md "Measurement"):
      gemm (compute dtype): 2 m n k per GEMM -- layer 1: forward + weight gradient (its input needs no gradient); layer 2
          (all n_sub rows) and layer 3 (seeds + hop 1 rows): forward, input gradient, weight gradient; the two G-sized
          MLP maps on the b seeds -- against the dense bf16 MFMA peak;
      gat edge: SURVEY 8(d)'s minimum per layer, 2 n H C s + E (8 + 2 H s) bytes, once forward and twice backward (target
          side and source side each read and write one n x H C image) -- against the 8 TB/s HBM peak (the family's
          microseconds include the aggregate-first last layer's own kernels, csrc/gat_tail.hip);
      optimizer: 8 fp32 streams over the flat buffers (p, g, m, v read; p, m, v written; g read once more for the norm);
      everything else is latency (20 workgroups of fp64 sweep, ~90 launches of a few microseconds)."""
    b0 = dd["dataloaders"][unit[0]][unit[1]]
    g1 = b0.graph
    lg = g1.layer_graphs
    H, C = cfg["gat_attention_heads"], cfg["gat_encoder_hidden"]
    HC = H * C
    s_el = 2 if compute_dtype == "bf16" else 4
    n, b = g1.n, b0.batch_size
    n1 = lg[1].n if lg is not None else n
    layers = [(n, n, g1.E)] + ([(lg[0].n, lg[0].n_tgt, lg[0].E), (lg[1].n, lg[1].n_tgt, lg[1].E)] if lg is not None
                              else [(n, n, g1.E), (n, n, g1.E)])
    hid_s, hid_d = cfg["svgp_encoder_layers"][0], cfg["decoder_layers"][-1]
    # layer 3: round 3 mapped all n1 = seeds + hop-1 rows; the aggregate-first form (csrc/gat_tail.hip, default since round 4)
    # maps the b aggregated seed rows -- the flops actually executed are what is priced, not the ones avoided
    tail_form = bool(lg is not None and lg[1].n_tgt * 4 <= lg[1].n and HC <= 2048)
    rows3 = b if tail_form else n1
    gemm_flops = (2 * 2.0 * n * G * HC            # layer 1: forward, weight gradient
                  + 3 * 2.0 * n * HC * HC          # layer 2
                  + 3 * 2.0 * rows3 * HC * HC      # layer 3
                  + 2 * 2.0 * b * G * hid_s        # SVGP encoder's G-sized map: forward, weight gradient
                  + 3 * 2.0 * b * hid_d * G)       # decoder's G-sized map
    gat_bytes = sum(3 * ((ns + nt) * HC * s_el + E * (8 + 2 * H * s_el)) for ns, nt, E in layers)
    opt_bytes = 8.0 * 4 * n_params
    out = {"workload": f"one step: n_sub={n}, seeds+hop1={n1}, b={b}, E={g1.E}, G={G}, H={H}, C={C}, {compute_dtype}, "
                       f"layer 3 {'aggregate-first (b rows mapped)' if tail_form else 'map-first (n1 rows mapped)'}",
           "live_ms_per_step": live_ms_per_step,
           "algorithmic": {"gemm_flops_per_step": gemm_flops, "gat_edge_bytes_per_step": gat_bytes, "optimizer_bytes_per_step": opt_bytes}}
    fam_path = newest_profile(f"train_{tag}_{compute_dtype}_families.json")
    if fam_path is None:
        out["source"] = "no committed rocprofv3 family summary for this dtype: only the algorithmic work is reported"
        return out
    prof = json.load(open(fam_path))
    fam = prof["families"]
    us = lambda *names: sum(fam.get(k, {}).get("us_per_step", 0.0) for k in names)
    gemm_us = us("gemm_bf16_library", "gemm_bf16_own") if compute_dtype == "bf16" else us("gemm_f32_library", "gemm_f32_own")
    gemm_peak = MFMA_BF16_PEAK_TFLOPS if compute_dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    gat_us, opt_us = us("gat_edge"), us("optimizer")
    out["source"] = (f"kernel microseconds REPLAYED from {os.path.relpath(fam_path, ROOT)} (rocprofv3 --kernel-trace of "
                     f"`bench.py --leg train`, {prof['steps']} steps, {prof.get('wall_us_per_step_without_profiler_stalls', prof['wall_us_per_step']):.0f} us/step under the profiler); "
                     "algorithmic flops / bytes computed live from this run's batch")
    out["profile_head"] = prof.get("head")
    out["profile_source_sha16"] = prof.get("source_sha16")
    out["stale_profile"] = prof.get("source_sha16") != source_fingerprint()      # True: the tree changed since it was profiled
    if "mfma" in prof:          # at most four kernels' MFMA-busy fractions travel in the line; the table stays in profiles/
        out["mfma_busy_top"] = mfma_top(prof["mfma"])
    out["families"] = {
        "gemm_" + compute_dtype: {"bound": "mfma", "us_per_step": gemm_us, "achieved": gemm_flops / max(gemm_us, 1e-9) / 1e6,
                                  "peak": gemm_peak, "unit": "TFLOP/s",
                                  "frac": gemm_flops / max(gemm_us, 1e-9) / 1e6 / gemm_peak},
        "gat_edge": {"bound": "hbm", "us_per_step": gat_us, "achieved": gat_bytes / max(gat_us, 1e-9) / 1e3, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": gat_bytes / max(gat_us, 1e-9) / 1e3 / HBM_PEAK_GBS},
        "optimizer": {"bound": "hbm", "us_per_step": opt_us, "achieved": opt_bytes / max(opt_us, 1e-9) / 1e3, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": opt_bytes / max(opt_us, 1e-9) / 1e3 / HBM_PEAK_GBS},
        "svgp_sweep": {"bound": "latency (2 L = 40 workgroups, one per matrix; fp64 vector FMA)", "us_per_step": us("svgp_sweep")},
        "gemm_f64_svgp": {"bound": "latency (m x m, b x m fp64 library GEMMs)", "us_per_step": us("gemm_f64_library")},
        "small_kernels": {"bound": "launch latency", "us_per_step": us("own_small", "torch_glue"),
                          "launches_per_step": sum(fam.get(k, {}).get("launches_per_step", 0.0) for k in ("own_small", "torch_glue")),
                          "torch_glue_us_per_step": us("torch_glue")},
    }
    out["launches_per_step"] = prof["launches_per_step"]
    out["under_12us"] = prof["under_12us"]
    return out


def epoch_block(tu, model, opt, cfg, dd, stepper, T, beta1, torch):
    """One whole training epoch as train_SpaDOT runs it (_train_utils.py:174-231): every batch of every time point
    (cfg3: 100 steps), then _update_Kmeans (per-time-point inference + K-means, kmeans_backend as configured) and
    _update_OT_matrix -- the per-epoch work the steps/s headline does not contain.  Seconds, host wall clock."""
    import random
    epoch = cfg["ot_epoch"]
    order = [(i, t) for i, t in enumerate(cfg["timepoints"]) if t in dd["dataloaders"]]
    nsteps = sum(len(dd["dataloaders"][t]) for _, t in order)

    def run_steps():
        with (stepper.chained() if stepper is not None else contextlib.nullcontext()):      # as train_SpaDOT's epoch loop does
            for tp_i, tp in order:
                for bi in range(len(dd["dataloaders"][tp])):
                    if stepper is not None:
                        stepper.step(tp_i, tp, bi, epoch, beta1)
                    else:
                        tu.training_step(model, opt, cfg, dd, tp_i, tp, bi, epoch, beta1)

    t0 = time.perf_counter()
    for _ in range(2 if stepper is not None else 0):      # first visits: eager, then capture (epochs 0 and 1 of a real run)
        run_steps()
        tu._update_Kmeans(model, cfg, dd)                 # (its phases are replayed as graphs from the second call on, too)
        tu._update_OT_matrix(model, cfg)
        model.train()
    torch.cuda.synchronize()
    warm_s = time.perf_counter() - t0
    random.shuffle(order)
    model.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tu._update_Kmeans(model, cfg, dd)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tu._update_OT_matrix(model, cfg)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    model.train()
    ot_every = int(cfg["ot_config"].get("ot_epochs", 10))
    return {"steps": nsteps, "steps_s": t1 - t0, "update_kmeans_s": t2 - t1, "update_ot_matrix_s": t3 - t2, "total_s": t3 - t0,
            "steps_per_s_whole_epoch": nsteps / (t3 - t0),
            "steps_per_s_reference_cadence": nsteps / ((t2 - t0) + (t3 - t2) / max(1, ot_every)),
            "kmeans_backend": cfg.get("kmeans_backend", "device"), "graph_warmup_s": warm_s,
            "note": "replayed-graph epoch (epoch >= 2 of a run): every (time point, batch) step, then the per-epoch K-means "
                    "refit of all time points (kmeans_backend 'device': its own k-means++ stream -- the fitted centres are not "
                    "sklearn's, the labels for given centres are) and the 10x10 OT plans.  steps_per_s_whole_epoch refreshes the "
                    f"plans EVERY epoch; steps_per_s_reference_cadence every {ot_every}th, as the reference does "
                    "(_train_utils.py:230-231)"}



# ------------------------------------------------------------------------------ the ONE stdout line

LINE_LIMIT = 6000      # the driver keeps the tail of stdout (r03: 8.6 KB); a longer line cannot be parsed
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def mfma_top(table, n=4):
    """The `n` kernels with the most MFMA-busy cycles per profiled run of a tools/mfma_summary.py table
    ({"kernels": [{"kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES", "mfma_util"}, ...]}), as {short name: mfma_util}."""
    items = []
    for r in (table.get("kernels", []) if isinstance(table, dict) else table):
        name = str(r.get("kernel", "?")).replace("void (anonymous namespace)::", "").split("(")[0]
        if name.startswith("Cijk_"):                                   # library GEMM: keep the macro-tile token
            mt = [t for t in name.split("_") if t.startswith("MT")]
            name = "lib_" + (mt[0] if mt else name[:24])
        w = float(r.get("launches", 1)) * float(r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0))
        if r.get("mfma_util") is not None:
            items.append((w, name[:40], float(r["mfma_util"])))
    items.sort(reverse=True)
    out = {}
    for _, nm, u in items:
        if nm not in out and len(out) < n:
            out[nm] = round(u, 3)
    return out


def _r(x, sig=6):
    """Floats to `sig` significant digits (recursively): the line is read by people and parsers, not re-used as input."""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, *keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_line(full, detail_file=None):
    """The record that goes to stdout: every key the contract names, the two extra objects (`roofline`, `cpu_baseline`)
    and one-number summaries of everything else; the full record (long `what` / `source` / `note` texts, per-repeat
    timings, per-loss errors, kernel tables) is written to `detail_file`.  Raises if the line would exceed LINE_LIMIT."""
    o = _pick(full, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "dtype_detail", "data")
    cfg = full.get("config", {})
    tr = cfg.get("train", {})
    o["config"] = {"workload": cfg.get("workload"), "parallelism": cfg.get("parallelism")}
    o["config"].update(_pick(tr, "n_sub", "E_sub", "m_inducing", "params", "hip_graphs", "staged_graphs",
                             "bucketed_grad_exchange", "units_in_schedule", "setup_s"))
    if "repeats" in tr:
        o["config"]["repeats_ms_per_step"] = _pick(tr["repeats"], "n", "min", "max", "spread_pct")
    o.update(_pick(full, "rccl_ranks", "rank_devices"))
    if isinstance(o.get("rank_devices"), list) and len(set(o["rank_devices"])) == 1:
        o["rank_devices"] = [o["rank_devices"][0]]                      # identical GPUs: name it once
    roof = full.get("roofline")
    if roof is not None:
        o["roofline"] = _pick(roof, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch",
                              "profile_kernel_us", "stale_profile")
        if "kernel_ms" in roof and "kernel" in roof:
            o["roofline"]["kernel_us_events"] = 1e3 * roof["kernel_ms"].get(roof["kernel"][2:], float("nan"))
        if "traffic_source" in roof:
            o["roofline"]["traffic_source"] = ("REJECTED (below algorithmic bytes)" if roof["traffic_source"].startswith("REJECTED")
                                               else "replayed from the committed --pmc summary in profiles/")
    rt = full.get("roofline_train")
    if rt is not None:
        fams = {}
        for k, v in rt.get("families", {}).items():
            fams[k] = _pick(v, "us_per_step", "frac", "launches_per_step")
            if "bound" in v:
                fams[k]["bound"] = v["bound"].split(" ")[0]
        o["roofline_train"] = {"families": fams}
        o["roofline_train"].update(_pick(rt, "launches_per_step", "stale_profile", "mfma_busy_top"))
        if "under_12us" in rt:
            o["roofline_train"]["under_12us"] = rt["under_12us"]
        if "algorithmic" in rt:
            o["roofline_train"]["algorithmic"] = rt["algorithmic"]
    if "epoch" in full:
        o["epoch"] = _pick(full["epoch"], "steps", "steps_s", "update_kmeans_s", "update_ot_matrix_s", "steps_per_s_whole_epoch",
                           "steps_per_s_reference_cadence", "kmeans_backend")
    if "parity_check" in full:
        o["parity_check"] = _pick(full["parity_check"], "max_rel_loss_err", "latent_rel_l2_err", "grad_cos_min", "grad_cos_min_param",
                                  "grad_rel_l2_max", "grad_cos_global")
    if "sinkhorn" in full:
        o["sinkhorn"] = _pick(full["sinkhorn"], "value", "unit", "ms_per_iter", "problem", "storage", "full_solve_s", "full_solve_iters",
                              "iters_per_s_with_convergence_checks", "pair_end_to_end_s", "cost_setup_s")
        if "full_solve_s" in o["sinkhorn"] and o["sinkhorn"].get("full_solve_iters"):
            o["sinkhorn"]["iters_per_s_whole_solve"] = o["sinkhorn"]["full_solve_iters"] / o["sinkhorn"]["full_solve_s"]
        o["sinkhorn"]["measurement_changed_r04"] = True       # raw rate without a host sync in the region (r01-r03 had one per 10)
    if "parity_check_sinkhorn" in full:
        o["parity_check_sinkhorn"] = _pick(full["parity_check_sinkhorn"], "stage_iters_equal", "stage_iters_max_diff", "marginal_rel_err",
                                           "plan_rel_err_top", "entries_compared", "oracle_solve_s")
    for k in ("cpu_baseline", "cpu_baseline_sinkhorn"):
        if k in full:
            o[k] = dict(_pick(full[k], "value", "unit", "cores", "kind"), sample=str(full[k].get("sample", ""))[:200])
    if detail_file:
        o["detail_file"] = detail_file
    o = _r(o)
    line = json.dumps(o, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:
        raise RuntimeError(f"bench line is {len(line)} characters (limit {LINE_LIMIT}): move detail into the side file")
    return o, line


def write_detail(full, n_gpus):
    """The full record as a side file (gpurun_out/ when the repo is writable, else the temp dir); returns its path or None."""
    import tempfile
    for base in (os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(base, exist_ok=True)
            p = os.path.join(base, f"bench_detail_{n_gpus}gpu.json")
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            return os.path.relpath(p, ROOT) if p.startswith(ROOT) else p
        except OSError:
            continue
    return None


# ------------------------------------------------------------------------------ main

# BASELINE.json's single-GPU shapes (cfg2 = configs[1], cfg3 = configs[2], cfg5shape = configs[4]'s per-GPU
# shape: two of its ten time points, the 256 inducing points scaled to them).  The driver's default command is cfg3.
PRESETS = {
    "cfg3": dict(spots=10000, genes=3000, timepoints=5, compute_dtype="bf16", inducing=1200),
    "cfg2": dict(spots=5000, genes=2000, timepoints=2, compute_dtype="f32", inducing=1200),
    "cfg5shape": dict(spots=20000, genes=5000, timepoints=2, compute_dtype="bf16", inducing=52),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spots", type=int, default=None, help="spots per time point (N_t)")
    ap.add_argument("--genes", type=int, default=None)
    ap.add_argument("--timepoints", type=int, default=None)
    ap.add_argument("--compute-dtype", default=None, choices=["bf16", "f32"])
    ap.add_argument("--ot-storage", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = their median")
    ap.add_argument("--cpu-steps", type=int, default=3, help="oracle training steps timed for cpu_baseline")
    ap.add_argument("--no-epoch", action="store_true", help="skip the whole-epoch block (100 steps + K-means + OT update)")
    ap.add_argument("--leg", default="both", choices=["both", "train", "sinkhorn"])
    ap.add_argument("--no-sinkhorn-parity", action="store_true",
                    help="skip the whole-solve oracle check of the Sinkhorn leg (about a minute of one host thread)")
    ap.add_argument("--preset", default="cfg3", choices=sorted(PRESETS),
                    help="BASELINE.json shape: cfg3 (default; the one `metric` is quoted on), cfg2 (configs[1]: 2 x 5000 x 2000, fp32, "
                         "m ~ 600) or cfg5shape (configs[4] per GPU: 20000 x 5000 per time point, m ~ 26, bf16).  Explicit "
                         "--spots / --genes / --timepoints / --compute-dtype / --inducing override the preset's values")
    ap.add_argument("--inducing", type=int, default=None, help="inducing_point_nums over all time points (config.yaml: 1200)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher rehearsal: start the ranks, run the all-reduce-of-ones proof on the chosen backend, print the "
                         "line's header and exit -- no GPU is touched (tests/test_bench_line_cpu.py, SPADOT_BENCH_BACKEND=gloo)")
    args = ap.parse_args(argv)
    pre = PRESETS[args.preset]
    args.custom = False
    for k in ("spots", "genes", "timepoints", "compute_dtype", "inducing"):
        if getattr(args, k) is None:
            setattr(args, k, pre[k])
        elif getattr(args, k) != pre[k]:
            args.custom = True
    return args


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: THIS process has made no GPU call (torch
    is not even imported yet), so it may start the N ranks itself -- as children, `python -m torch.distributed.run`, the
    command the driver uses -- relay rank 0's one stdout line and leave with their exit code.  Never an exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] launching:", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and len(lines) != 1:
        print(f"[bench] expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 3
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    # stdout carries exactly ONE line (the JSON); everything the library prints goes to stderr
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    try:
        _main(real_stdout, args)
    finally:
        sys.stdout = real_stdout


def _rendezvous_only(args, real_stdout):
    """The launcher's rehearsal: process group on the chosen backend, the all-reduce-of-ones proof, header line."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("SPADOT_BENCH_BACKEND", "nccl")
    ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
        ones = torch.ones(1, device="cpu" if backend == "gloo" else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
        dist.all_reduce(ones)
        ranks = int(ones.item())
    ok = ranks == world == args.gpus
    if rank == 0:
        print(json.dumps({"metric": METRIC, "n_gpus": world, "rccl_ranks": ranks, "backend": backend, "rendezvous_only": True,
                          "value": None}), file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        print(f"[bench] world {world} / all-reduce of ones {ranks} / --gpus {args.gpus} disagree", file=sys.stderr)
        sys.exit(4)


def _main(real_stdout, args):
    if args.rendezvous_only:
        return _rendezvous_only(args, real_stdout)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: refusing to print a line whose "
              "n_gpus is not the one asked for", file=sys.stderr)
        sys.exit(4)

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

    # proof that the collectives ran over `world` ranks of the chosen backend: an all-reduce of ones
    rccl_ranks, rank_devices = 1, [torch.cuda.get_device_name(torch.cuda.current_device())]
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        if not (rccl_ranks == world == dist.get_world_size() == args.gpus):
            print(f"[bench] rank {rank}: all-reduce of ones = {rccl_ranks}, world {world}, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(4)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, f"{local_rank}:{torch.cuda.get_device_properties(local_rank).name}")

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
        # N == 1: the whole 5-time-point job on one GPU.  N > 1: the same job dealt to the ranks batch by batch
        # (spadot_amd.parallel, granularity 'batch': batch bi of time point t runs on rank (20 t + bi) mod N for the whole
        # run); every rank holds every time point's rows (0.3 GB) and K-means state, builds and caches only ITS batches;
        # the model is replicated, gradients are all-reduced.
        data = make_dataset(T, N, G, seed=1993)
        cfg.update(input_dim=G, timepoints=list(range(T)), device=torch.device(dev), compute_dtype=cdt,
                   inducing_point_nums=int(args.inducing))
        plan = None
        if world == 1:
            own = list(range(T))
            cfg["owned_timepoints"] = own
        else:
            from spadot_amd.parallel import configure_shard
            cfg["shard_granularity"] = "batch"
            plan = configure_shard(data, cfg, world, rank)
            own = list(cfg["owned_timepoints"])
        owned_all = [own]
        if world > 1:
            owned_all = [None] * world
            dist.all_gather_object(owned_all, own)
        _utils.set_seed(cfg["seed"])
        t_setup = time.perf_counter()
        dd = tu.prepare_dataloader(data, cfg)
        del data
        model = SpaDOT.SpaDOT(cfg, dd).to(dev)
        opt = FlatAdamW(model.parameters(), lr=cfg["lr"], last=model.GATEncoder.first_layer_parameters(),
                        first=model.SVGPEncoder.parameters())
        tu._update_Kmeans(model, cfg, dd)          # labels/centres so the K-means and OT terms are live
        tu._update_OT_matrix(model, cfg)
        setup_s = time.perf_counter() - t_setup
        model.train()
        grad_sync = grad_sync_async = None
        if world > 1:       # two buckets, same order on every rank and path (spadot_amd.parallel.make_grad_sync)
            from spadot_amd.parallel import make_grad_sync
            grad_sync, grad_sync_async = make_grad_sync(opt)
            opt.grad_scale.fill_(1.0 / world)       # every rank has a batch in every step: update on the MEAN gradient
        # schedule: (tp_i, batch) round robin over the time points that have a predecessor on this rank
        train_tps = [t for t in own if t >= 1 and (t - 1) in own]
        sched = [(t, bi) for bi in range(len(dd["dataloaders"][train_tps[0]])) for t in train_tps
                 if plan is None or plan.unit_owner(t, bi) == rank]
        assert sched, f"rank {rank} has no batch of a time point with a predecessor"
        epoch = cfg["ot_epoch"]                     # every loss term active
        beta1 = 0.5

        # Steady-state training replays one captured hipGraph per (time point, batch) (epochs >= 2 of a real
        # run); the keys the timed region touches are visited twice beforehand (eager, then capture).
        # Multi-rank: the same staged graphs; the flat gradient crosses RCCL in two buckets (not captured), the first
        # beside the backward pass of the first GAT layer, then one clip + AdamW graph.
        use_graphs = os.environ.get("SPADOT_BENCH_NO_GRAPHS") != "1"
        stepper = (tu.GraphedStepper(model, opt, cfg, dd, grad_sync=grad_sync, grad_sync_async=grad_sync_async)
                   if use_graphs else None)

        if stepper is not None:
            stepper.clone_output = False          # like train_SpaDOT: the losses are consumed in-stream
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
        chain = state["stepper"].chained if state["stepper"] is not None else contextlib.nullcontext
        with chain():                                # consecutive steps, like the inner loop of train_SpaDOT's epoch
            for i in range(args.warmup):
                step(i)
        # `repeats` timed regions of EXACTLY `steps` steps each, every one bracketed by barrier + synchronize and
        # reduced with MAX over ranks; `value` is the median region (spread reported beside it)
        regions = []
        for rep in range(max(1, args.repeats)):
            barrier()
            t0 = time.perf_counter()
            with chain():
                for i in range(args.steps):
                    last = step(args.warmup + i)
            barrier()
            regions.append(max_over_ranks(time.perf_counter() - t0))
        el = float(np.median(regions))
        b0 = dd["dataloaders"][sched[0][0]][sched[0][1]]
        train_res = {"value": world * args.steps / el, "ms_per_step": 1e3 * el / args.steps,
                     "repeats": {"n": len(regions), "ms_per_step": [1e3 * r / args.steps for r in regions],
                                 "min": 1e3 * min(regions) / args.steps, "max": 1e3 * max(regions) / args.steps,
                                 "spread_pct": 100.0 * (max(regions) - min(regions)) / el},
                     "setup_s": setup_s, "n_sub": b0.graph.n, "E_sub": b0.graph.E,
                     "m_inducing": int(dd["inducing_points"][sched[0][0]].shape[0]),
                     "params": int(opt.count), "last_losses": [float(v) for v in last.cpu().tolist()],
                     "hip_graphs": state["stepper"] is not None,
                     "staged_graphs": bool(state["stepper"] is not None and state["stepper"].staged),
                     "bucketed_grad_exchange": bool(state["stepper"] is not None and state["stepper"].overlap)}
        train_res["units_in_schedule"] = len(sched)
        train_res["roofline_train"] = roofline_train(dd, sched[0], cfg, G, args.compute_dtype, int(opt.count),
                                                     train_res["ms_per_step"], tag="custom" if args.custom else args.preset)
        if world == 1 and not args.no_epoch:
            train_res["epoch"] = epoch_block(tu, model, opt, cfg, dd, state["stepper"], T, beta1, torch)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            t, bi = sched[0]
            train_res["cpu_baseline"], train_res["parity_check"] = cpu_train_steps(
                model, opt, dd, cfg, tu, t, bi, t - 1, epoch, beta1, n_steps=max(1, args.cpu_steps))
        del model, opt, dd
        torch.cuda.empty_cache()

    # ========================================================================== Sinkhorn leg
    sk_res, roof = None, None
    if args.leg in ("both", "sinkhorn"):
        from spadot_amd.ot import OTSolver
        I = J = N
        solver = OTSolver(I, J, storage=args.ot_storage, device=dev)
        lat_x, lat_y = synthetic_latents(I, 100 + rank), synthetic_latents(J, 200 + rank)
        solver.set_cost_from_latents(lat_x, lat_y)
        t0 = time.perf_counter()
        info = solver.solve(OT_CFG)                 # whole 6-stage solve (first call: code objects load, attributes are set)
        torch.cuda.synchronize()
        solve_cold_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        info2 = solver.solve(OT_CFG)                # the same solve again from scratch; leaves a converged state
        torch.cuda.synchronize()
        solve_s = time.perf_counter() - t0
        assert list(info2.stage_iters) == list(info.stage_iters)
        # one pair end to end, as compute_transport_map runs it (ot_solvers.py:95-121): cost matrix from the latents (sqeuclidean /
        # exact median), the six-stage solve, the plan R / J left in HBM -- what full_solve_s leaves out is the cost setup
        lx, ly = torch.as_tensor(lat_x, device=dev), torch.as_tensor(lat_y, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.set_cost_from_latents(lx, ly)
        torch.cuda.synchronize()
        cost_setup_s = time.perf_counter() - t0
        solver.solve(OT_CFG)
        plan_dev = solver.plan("torch")
        torch.cuda.synchronize()
        pair_e2e_s = time.perf_counter() - t0
        del plan_dev, lx, ly
        want_parity = rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_sinkhorn_parity
        # the solve's plan for the oracle check: taken NOW (before the timed iterations move a, b on) but kept on the device;
        # its 800 MB copy to pageable host memory follows the timed region (round 4: with that copy in front of it the region's
        # 20 x 10 iterations took 82.4 us each by the wall clock where the same 200 iterations between two HIP events take 75.9)
        plan_first_dev = solver.plan("torch", dtype=torch.float64) if want_parity else None
        for _ in range(args.warmup):
            solver.run_iterations(OT_CFG, OT_CFG["epsilon"], ITERS_PER_STEP, timed=False)
        # the timed region: EXACTLY `steps` steps of ITERS_PER_STEP iterations, barrier + synchronize on both sides and nothing
        # that waits for the device in between (until round 4 every step read its own HIP events back: a host
        # synchronisation per 10 iterations, 9 us per iteration of drained queue that no solve pays -- the solver's own loop
        # reads one 16-byte record per 15 iterations, `iters_per_s_with_convergence_checks` below)
        solver.tau_flag(reset=True)                 # (clears what the solves and the warm-up may have left)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            solver.run_iterations(OT_CFG, OT_CFG["epsilon"], ITERS_PER_STEP, timed=False)
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        # the untimed launches read nothing back: the tau flag they accumulate is read ONCE, behind the closing barrier -- a run
        # in which a scaling exceeded tau is not the steady-state schedule a solve executes
        if solver.tau_flag(reset=True):
            raise RuntimeError("a scaling exceeded tau inside the timed Sinkhorn region: not a steady-state timing run")
        iters = args.steps * ITERS_PER_STEP
        ev_ms = solver.run_iterations(OT_CFG, OT_CFG["epsilon"], iters, timed=True)      # the same iterations between two HIP events
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
                  "full_solve_s": solve_s, "full_solve_first_call_s": solve_cold_s, "full_solve_iters": int(sum(info.stage_iters)),
                  "cost_setup_s": cost_setup_s, "pair_end_to_end_s": pair_e2e_s,
                  "iters_per_s_with_convergence_checks": ck_it / (ck_ms * 1e-3),
                  # `value` is the raw launch rate (no convergence measure); r01-r03 reported a region with a host
                  # synchronisation every 10 iterations, so their 11.7-12.6 k are not like-for-like with r04+
                  "value_is": "raw rate since r04 (no host sync inside the region); in-solve rate = iters_per_s_with_convergence_checks"}
        roof = {"bound": "hbm", "kernel": "k_" + dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": alg,
                "kernel_ms": kt, "fused_geometry": geo}
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2 for 16-byte/lane
        # streams on gfx950, + WRITE_SIZE; MI355X_MICROARCH.md "HBM"): taken from the committed summary of the
        # same workload, when there is one
        pmc = newest_profile("sinkhorn_cfg3_f32_pmc_summary.csv")
        if dom == "fused_pass" and I == 10000 and args.ot_storage == "f32" and pmc:
            import csv
            for row in csv.reader(open(pmc)):
                if row and "k_fused_pass<float, 5, 2" in row[0]:
                    traffic = (2.0 * float(row[1]) + float(row[3])) * 1024.0
                    src = (f"replayed from {os.path.relpath(pmc, ROOT)} (separate rocprofv3 --pmc passes of this command with "
                           "SPADOT_OT_SPEC_BATCHES=1, i.e. no launch that returns on the stop word: 2*FETCH_SIZE + WRITE_SIZE, "
                           "bytes per launch) -- not measured in this run")
                    if traffic >= alg:
                        roof["traffic"], roof["traffic_source"] = traffic, src
                    else:       # a kernel cannot move less than it must read: such a mean is polluted, not a result
                        roof["traffic_source"] = (f"REJECTED: {src}; its mean of {traffic:.0f} bytes per launch is below the "
                                                  f"{alg:.0f} algorithmic bytes (launches that did not sweep are in it)")
            meta = newest_profile("sinkhorn_cfg3_f32_profile_meta.json")
            if meta:
                m = json.load(open(meta))
                roof["profile_kernel_us"] = m.get("fused_pass_avg_us")
                roof["profile_source_sha16"] = m.get("source_sha16")
                roof["stale_profile"] = m.get("source_sha16") != source_fingerprint()
        solver.close()
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            sk_res["cpu_baseline"] = cpu_sinkhorn(I, J, budget_s=6.0 if want_parity else 10.0)
            if want_parity:
                plan_first = plan_first_dev.cpu().numpy()
                del plan_first_dev
                sk_res["parity_check"] = cpu_sinkhorn_parity(lat_x, lat_y, plan_first, info.stage_iters)
                del plan_first

    if rank == 0:
        if train_res is not None:
            out.update(value=train_res["value"], unit="training steps/s", ms_per_step=train_res["ms_per_step"])
        else:
            out.update(value=sk_res["value"], unit="Sinkhorn iters/s", ms_per_step=sk_res["ms_per_iter"] * ITERS_PER_STEP)
        out["dtype"] = args.compute_dtype
        out["dtype_detail"] = (f"{args.compute_dtype} GAT branch + linears, f32 parameters/optimizer, f64 SVGP algebra; "
                               f"Sinkhorn {args.ot_storage} kernel matrix with f64 scalings")
        name = "custom shape" if args.custom else args.preset
        out["config"] = {"workload": f"{name}: {T} time points x {N} spots x {G} genes, batch 512, k=30, {args.inducing} inducing points; "
                                     f"Sinkhorn pair problem {N}x{N}",
                         "train": {k: v for k, v in (train_res or {}).items()
                                   if k not in ("cpu_baseline", "parity_check", "roofline_train", "epoch")},
                         "parallelism": "1 GPU" if world == 1 else (
                             f"dp{world}: (time point, batch) units dealt round-robin to {world} ranks (one per GPU, backend "
                             f"{backend}), one batch per rank per step, flat-gradient all-reduce (sum, update on the mean) in 2 "
                             f"buckets, the first beside the backward pass. Per-rank replication cost: every rank holds all {T} "
                             f"time points' rows ({T * N * G * (2 if args.compute_dtype == 'bf16' else 4) / 1e9:.2f} GB) and, per epoch, "
                             f"refits all {T} K-means and solves all {T - 1} centre-pair OT problems redundantly (no broadcast); "
                             f"Sinkhorn leg: rank r solves its own {N}x{N} pair, no collective")}
        out["rccl_ranks"] = rccl_ranks
        if train_res is not None:
            out["rank_timepoints"] = owned_all
        out["rank_devices"] = rank_devices
        for k in ("roofline_train", "epoch", "parity_check"):
            if train_res is not None and k in train_res:
                out[k] = train_res[k]
        if sk_res is not None:
            out["sinkhorn"] = {k: v for k, v in sk_res.items() if k not in ("cpu_baseline", "parity_check")}
            if "parity_check" in sk_res:
                out["parity_check_sinkhorn"] = sk_res["parity_check"]
            out["roofline"] = roof
        cb = (train_res or {}).get("cpu_baseline") or (sk_res or {}).get("cpu_baseline")
        if cb:
            out["cpu_baseline"] = cb
            if train_res and sk_res and "cpu_baseline" in sk_res and "cpu_baseline" in train_res:
                out["cpu_baseline_sinkhorn"] = sk_res["cpu_baseline"]
        _, line = compact_line(out, write_detail(out, world))
        print(line, file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
