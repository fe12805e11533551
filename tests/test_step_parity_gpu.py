"""GPU: the training step at BASELINE.json's shapes against the fp64 host oracle.

  * cfg3 (the benchmarked shape: one time-point pair of 10 000 spots x 3 000 genes, batches of 512 seeds whose
    two-hop closure is ~the whole time point, ~240 inducing points): ONE optimizer step's arithmetic --
    the seven loss terms, final_latent and the gradient of every parameter -- in bf16 compute (what bench.py times)
    and in fp32 compute, against oracle/model_oracle.training_step (the reference's formulas,
    _train_utils.py:187-217 -> SpaDOT.py:52-94, (b, m, m) ELBO tensor and all) on the same weights, batch, graph,
    noise, K-means state and OT plan.
  * cfg2 (2 x 5 000 spots x 2 000 genes, fp32, 1 200 inducing points -> m ~ 600): graphed steps == eager steps and
    one oracle-checked step: loss terms, latent, and the gradients of every GAT-encoder / decoder parameter (the SVGP
    branch without a tape on the host: its taped (b, m, m) tensors would need 15 GB).
  * cfg5 shape (20 000 spots x 5 000 genes per time point, 256 inducing points over 10 time points -> m ~ 26, k = 30):
    staged replay == eager, and one WHOLE oracle step (m ~ 26 makes it affordable) at cfg3's bf16 thresholds.

  * cfg1 shape (ChickenHeart-like ragged pair: 747 + 1 966 spots, k = 6 / 12, G = 500, last batches of 235 and 430
    seeds): one oracle-checked fp32 step per time point on its PARTIAL last batch (N_train / b with the partial b,
    svgp.py:63,74; per-time-point k, _train_utils.py:69-70).

Stated tolerances (the reference is fp64 on the CPU), asserted at about 3x what the MI355X measures:
  fp32 compute: loss terms rtol 1e-5, latent rtol 1e-5 / atol 1e-5 x scale, per-parameter gradient cosine >= 0.999999 and
                relative L2 error <= 2e-4 (measured: loss terms 1.1e-7, latent 3e-7, worst parameter relative L2 4.1e-5);
  bf16 compute (GAT branch and the two G-sized linears in bf16, fp32 accumulate): loss terms rtol 3e-4, latent relative
                L2 <= 3e-3, gradient as a whole (the direction AdamW follows) cosine >= 0.99999; per parameter: the
                attention vectors att_src / att_dst (2 048 numbers each, sums of bf16 products over ~10^4 nodes) cosine
                >= 0.9985 and relative L2 <= 0.07 (round 4: tightened from 0.998 / 0.1), every other parameter cosine >= 0.9998
                and relative L2 <= 0.03.
                (Measured, round 2 final tree: loss terms 2.1e-5, latent 8.5e-4, whole gradient 0.999998; attention
                vectors: worst cosine 0.99941 / relative L2 0.049 (gat3.att_dst); every other parameter: cosine >= 0.99996,
                relative L2 <= 0.0087.)
  The same thresholds hold for the step as bench.py runs it -- the REPLAYED staged hipGraphs, fed the same noise.
"""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(T, N, G, n_ind, compute_dtype, bf16_exact_inputs=False, seed=1993, **over):
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(T, N, G, seed=seed)
    if bf16_exact_inputs:       # expression values representable in bf16: the fp32 and bf16 runs read the same numbers
        data.X = torch.from_numpy(data.X).bfloat16().float().numpy()
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=G, timepoints=list(range(T)), device=torch.device(DEV), compute_dtype=compute_dtype,
               inducing_point_nums=n_ind, kmeans_backend="sklearn")
    cfg.update(over)
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    del data
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    return tu, cfg, dd, model, opt


def _report(tag, rep):
    """Keeps the numbers of the last run next to the other GPU artefacts (gpurun_out/ travels back from the box)."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"step_parity_{tag}.json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_cfg3_training_step_matches_the_fp64_oracle(dtype):
    from oracle import step_parity as sp
    cdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    tu, cfg, dd, model, opt = _setup(2, 10000, 3000, 480, cdt, bf16_exact_inputs=True)
    tp, bi, epoch, beta1 = 1, 0, cfg["ot_epoch"], 0.5          # every loss term active
    batch = dd["dataloaders"][tp][bi]
    assert batch.graph.n > 9000 and batch.graph.E > 250000 and batch.batch_size == 512
    noise = sp.make_noise(batch.batch_size, seed=0)
    # the oracle's answer depends on the weights (same seed -> same initial weights in both dtypes), the data (bf16-exact
    # here) and the K-means / OT state (fitted on the device latents, so it differs slightly between the two runs):
    # one oracle step per run, ~30 s on the GPU box's cores
    inp = sp.oracle_inputs(model, dd, cfg, tp, bi, tp - 1)
    assert 200 <= inp["m"] <= 280
    ref = sp.oracle_step(inp, cfg, beta1, noise)
    dl, dz, dg = sp.device_step(model, opt, cfg, dd, tu, 1, tp, bi, epoch, beta1, noise)
    rep = sp.compare(dl, dz, dg, ref)
    rep["dtype"], rep["oracle_seconds"] = dtype, ref["seconds"]
    _report(f"cfg3_{dtype}", rep)
    print(json.dumps({k: v for k, v in rep.items() if k != "per_param"}))
    assert np.isfinite(dl).all() and all(np.isfinite(v).all() for v in dg.values())
    assert (np.asarray(rep["loss_ref"])[4:] > 0).all()         # alignment, K-means and OT terms are live
    _assert_step_parity(rep, dtype, ref)

    # the step as bench.py runs it: the same batch through the REPLAYED staged graphs (eager visit, capture + replay,
    # replay), the same noise at a fixed address -- compared with the same oracle step, no eager hop in between
    model.fixed_noise = (noise[0].to(DEV), noise[1].float().to(DEV))
    stepper = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    stepper.keep_latents = True
    for k in range(3):
        opt.flat_grad.fill_(7.0)
        lr = stepper.fb(1, tp, bi, epoch, beta1)
    torch.cuda.synchronize()
    assert stepper.staged and len(stepper.graphs) == 1
    gr = {n: p.grad.detach().float().cpu().numpy().copy() for n, p in model.named_parameters()}
    rep2 = sp.compare(lr.detach().float().cpu().numpy().astype(np.float64), stepper.latents[(tp, bi)].float().cpu().numpy(), gr, ref)
    rep2["dtype"] = dtype
    _report(f"cfg3_{dtype}_replayed", rep2)
    print(json.dumps({k: v for k, v in rep2.items() if k != "per_param"}))
    _assert_step_parity(rep2, dtype, ref)


def _assert_step_parity(rep, dtype, ref):
    if dtype == "f32":
        assert rep["max_rel_loss_err"] <= 1e-5, rep["loss_rel_err"]
        assert rep["latent_max_abs_err"] <= 1e-5 * max(1.0, float(np.abs(ref["latent"]).max())) + 1e-5
        assert rep["grad_cos_min"] >= 0.999999, (rep["grad_cos_min_param"], rep["grad_cos_min"])
        assert rep["grad_rel_l2_max"] <= 2e-4, (rep["grad_rel_l2_max_param"], rep["grad_rel_l2_max"])
    else:
        assert rep["max_rel_loss_err"] <= 3e-4, rep["loss_rel_err"]
        assert rep["latent_rel_l2_err"] <= 3e-3
        assert rep["grad_cos_global"] >= 0.99999
        for name, (l2, cos) in rep["per_param"].items():
            if ".att_" in name:
                assert cos >= 0.9985 and l2 <= 0.07, (name, l2, cos)
            else:
                assert cos >= 0.9998 and l2 <= 0.03, (name, l2, cos)
    # a Linear bias in front of BatchNorm has a zero gradient in exact arithmetic: zero on the device as well
    assert len(rep["zero_grad_params"]) >= 2 and rep["zero_grad_dev_rel_norm_max"] <= 1e-6


def test_cfg2_fp32_graphed_steps_match_eager_and_the_oracle_step():
    """BASELINE.json configs[1]: 2 time points x 5 000 spots x 2 000 genes, fp32 compute, the default 1 200 inducing
    points (m ~ 600 per time point: the blocked SPD inverse), k = 30."""
    from oracle import model_oracle as mo, step_parity as sp
    tu, cfg, dd, model, opt = _setup(2, 5000, 2000, 1200, torch.float32)
    m = int(dd["inducing_points"][1].shape[0])
    assert 520 <= m <= 680
    tp, epoch, beta1 = 1, cfg["ot_epoch"], 0.5
    b0 = dd["dataloaders"][tp][0]
    assert b0.graph.n > 4500

    # (a) one step against the oracle: the seven loss terms, the latent, and the gradient of every GAT-encoder and decoder
    # parameter (round 5).  The SVGP branch is evaluated without a tape on the host -- its (b, m, m) tensor is 1.5 GB per
    # latent dimension at m ~ 600, ten of them alive under autograd -- which leaves the other parameters' gradients exact
    # (oracle/step_parity.oracle_step_without_svgp_tape); the SVGP encoder's gradients are checked at cfg3 / cfg1 / cfg5 sizes.
    noise = sp.make_noise(b0.batch_size, seed=3)
    inp = sp.oracle_inputs(model, dd, cfg, tp, 0, tp - 1)
    ref = sp.oracle_step_without_svgp_tape(inp, cfg, beta1, noise)
    dl, dz, dg = sp.device_step(model, opt, cfg, dd, tu, 1, tp, 0, epoch, beta1, noise)
    np.testing.assert_allclose(dl, ref["losses"], rtol=1e-4, err_msg=str(sp.LOSS_NAMES))
    np.testing.assert_allclose(dz, ref["latent"], rtol=1e-4, atol=1e-4)
    assert all(np.isfinite(v).all() for v in dg.values())
    rep = sp.compare(dl, dz, dg, ref)
    rep["oracle_seconds"] = ref["seconds"]
    _report("cfg2_f32", rep)
    print(json.dumps({k: v for k, v in rep.items() if k != "per_param"}))
    assert any(k.startswith("GATEncoder.gat1") for k in rep["per_param"]) and any(k.startswith("decoder.") for k in rep["per_param"])
    assert rep["max_rel_loss_err"] <= 1e-5, rep["loss_rel_err"]
    # fp32 thresholds of cfg3 (cosine 0.999999, relative L2 2e-4; measured here: <= 6.2e-5) for every parameter but the LAST
    # layer's attention vectors, which sit at 1.8e-3 / 5.9e-4 at this shape (cosine 0.9999985): their gradient is W_h (dS^T x)
    # with dS the softmax backward's row sums -- sums that vanish identically wherever a target's logits share a sign, so
    # fp32 rounding is seen relative to a cancelled total.  The kernel itself is at 2e-6 on random inputs of this shape
    # (profiles/r05/tail_f32_diag.txt); asserted at about 3x the measured values.
    for name, (l2, cos) in rep["per_param"].items():
        if name.startswith("GATEncoder.gat3.att_"):
            assert cos >= 0.99999 and l2 <= 5e-3, (name, l2, cos)
        else:
            assert cos >= 0.999999 and l2 <= 2e-4, (name, l2, cos)

    # (b) replayed graphs == eager steps on the same batches (noise silenced in both)
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    staged = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    assert staged.capturable
    for rep in range(3):                                         # eager, capture + replay, replay
        for bi in (0, 3):
            staged.beta1_t[1].fill_(-beta1)
            la = tu.forward_backward(model, cfg, dd, 1, tp, bi, epoch, staged.beta1_t, optimizer=opt)
            ga = opt.flat_grad.clone()
            opt.flat_grad.fill_(7.0)
            lb = staged.fb(1, tp, bi, epoch, beta1)
            gb = opt.flat_grad
            assert torch.isfinite(gb).all()
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-4, atol=1e-5)
            scale = float(ga.abs().max())
            np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=2e-3, atol=2e-4 * scale)
    assert len(staged.graphs) == 2
    # and a few real optimizer steps through the replayed graphs stay finite and move the parameters
    p0 = opt.flat_param.clone()
    for k in range(4):
        out = staged.step(1, tp, k % 2 * 3, epoch, beta1)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(opt.flat_param).all()
    assert float((opt.flat_param - p0).abs().max()) > 0


def test_cfg5_shape_step_runs_staged_equals_eager():
    """BASELINE.json configs[4] per-GPU shape: 20 000 spots x 5 000 genes per time point, 256 inducing points over 10
    time points (here 2 time points with 52 -> m ~ 26 each), k = 30, bf16 compute.  Batches gather their rows per step
    (batch_cache_gb = 0: at 40 batches x 20k rows x 5k genes the cache would be 8 GB per time point -- fine on the
    device, just not worth filling for three batches)."""
    tu, cfg, dd, model, opt = _setup(2, 20000, 5000, 52, torch.bfloat16, batch_cache_gb=0.0, kmeans_backend="device")
    m = int(dd["inducing_points"][1].shape[0])
    assert 15 <= m <= 40
    tp, epoch, beta1 = 1, cfg["ot_epoch"], 0.5
    b0 = dd["dataloaders"][tp][0]
    assert b0.y is None and b0.graph.n > 18000 and b0.graph.E > 500000
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    staged = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    for rep in range(3):
        for bi in (0, 39):                                        # first and last (partial) batch
            staged.beta1_t[1].fill_(-beta1)
            la = tu.forward_backward(model, cfg, dd, 1, tp, bi, epoch, staged.beta1_t, optimizer=opt)
            ga = opt.flat_grad.clone()
            opt.flat_grad.fill_(7.0)
            lb = staged.fb(1, tp, bi, epoch, beta1)
            gb = opt.flat_grad
            assert torch.isfinite(lb).all() and torch.isfinite(gb).all()
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-4, atol=1e-5)
            scale = float(ga.abs().max())
            np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=2e-3, atol=2e-4 * scale)
    la = la.cpu().numpy()
    assert (la[[1, 3, 4, 5, 6]] > 0).all()                      # Recon, GAT_KL, alignment, K-means, OT are live
    # one WHOLE oracle step at this shape (round 5; m ~ 26: the (b, m, m) ELBO tensor is 2.8 MB, the cost is the fp64 GAT
    # over 20 000 x 5 000): 7 loss terms, latent, every gradient, at cfg3's bf16 thresholds
    from oracle import step_parity as sp
    model.fixed_noise = None
    noise = sp.make_noise(b0.batch_size, seed=11)
    inp = sp.oracle_inputs(model, dd, cfg, tp, 0, tp - 1)
    ref = sp.oracle_step(inp, cfg, beta1, noise)
    dl, dz, dg = sp.device_step(model, opt, cfg, dd, tu, 1, tp, 0, epoch, beta1, noise)
    rep = sp.compare(dl, dz, dg, ref)
    rep["oracle_seconds"] = ref["seconds"]
    _report("cfg5shape_bf16", rep)
    print(json.dumps({k: v for k, v in rep.items() if k != "per_param"}))
    _assert_step_parity(rep, "bf16", ref)
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    for k in range(3):
        out = staged.step(1, tp, 0, epoch, beta1)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(opt.flat_param).all()


def test_cfg1_ragged_timepoints_partial_last_batches_match_the_oracle():
    """BASELINE.json configs[0] shape (the reference's CPU-runnable plumbing case, ChickenHeart's first two time points):
    747 + 1 966 spots -> k = 6 and 12, batches of 512 with PARTIAL last batches of 235 and 430 seeds, 500 genes, fp32
    compute.  One whole step (7 loss terms, latent, every gradient) per time point on its partial last batch against the
    fp64 oracle: N_train / b uses the partial b (svgp.py:63,74), the OT term is live for the second time point only."""
    from oracle import step_parity as sp
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, [747, 1966], 500, seed=7)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=500, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.float32,
               inducing_point_nums=480, kmeans_backend="sklearn")
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    assert [len(dd["dataloaders"][t]) for t in (0, 1)] == [2, 4]
    assert [dd["dataloaders"][t][-1].batch_size for t in (0, 1)] == [235, 430]
    deg = [int(dd["graphs"][t].E) // dd["graphs"][t].n - 1 for t in (0, 1)]
    assert deg == [6, 12], deg                                   # k = min(30, 6 * round(N_t / 1000)) per time point
    assert sum(int(dd["inducing_points"][t].shape[0]) for t in (0, 1)) == 480
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    epoch, beta1 = cfg["ot_epoch"], 0.5
    for tp_i, tp in enumerate((0, 1)):
        bi = len(dd["dataloaders"][tp]) - 1
        b = dd["dataloaders"][tp][bi].batch_size
        noise = sp.make_noise(b, seed=tp)
        inp = sp.oracle_inputs(model, dd, cfg, tp, bi, tp - 1, do_km=True, do_ot=tp_i != 0)
        assert inp["b"] == b
        ref = sp.oracle_step(inp, cfg, beta1, noise)
        dl, dz, dg = sp.device_step(model, opt, cfg, dd, tu, tp_i, tp, bi, epoch, beta1, noise)
        rep = sp.compare(dl, dz, dg, ref)
        _report(f"cfg1_tp{tp}_f32", rep)
        print(json.dumps({k: v for k, v in rep.items() if k != "per_param"}))
        assert (rep["loss_ref"][6] > 0) == (tp_i != 0)           # OT term: only with a predecessor
        assert rep["loss_ref"][5] > 0
        assert rep["max_rel_loss_err"] <= 1e-4, rep["loss_rel_err"]
        assert rep["latent_max_abs_err"] <= 1e-4 * max(1.0, float(np.abs(ref["latent"]).max())) + 1e-5
        assert rep["grad_cos_min"] >= 0.9999, (rep["grad_cos_min_param"], rep["grad_cos_min"])
        assert rep["grad_rel_l2_max"] <= 2e-3, (rep["grad_rel_l2_max_param"], rep["grad_rel_l2_max"])
    # and the staged replay of the partial batch gives what its eager step gives
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    staged = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    for rep_i in range(3):
        staged.beta1_t[1].fill_(-beta1)
        la = tu.forward_backward(model, cfg, dd, 1, 1, 3, epoch, staged.beta1_t, optimizer=opt)
        ga = opt.flat_grad.clone()
        opt.flat_grad.fill_(7.0)
        lb = staged.fb(1, 1, 3, epoch, beta1)
        np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(opt.flat_grad.cpu().numpy(), ga.cpu().numpy(), rtol=2e-3, atol=2e-4 * float(ga.abs().max()))


def test_cfg4_width_step_staged_equals_eager_and_the_oracle_forward():
    """BASELINE.json configs[3] width: MouseOrganogenesis has 9 281 SVGs (examples/MouseOrganogenesis_output/SVG_genes.txt of the
    reference); the gene axis is padded to 9 344 in the cached batch rows.  (a) bf16, N_t = 10 000: the replayed staged step
    equals the eager step, everything finite -- the K = 9 344 paths of the dense maps (library forward, both weight-gradient
    tilings and their 4 GiB guards, the G-sized decoder map and k_bias_sqerr_* at G > 5 000).  (b) fp32, N_t = 2 000: one
    forward against the fp64 oracle (7 loss terms, latent)."""
    from oracle import model_oracle as mo, step_parity as sp
    G = 9281
    tu, cfg, dd, model, opt = _setup(2, 10000, G, 480, torch.bfloat16, kmeans_backend="device")
    tp, epoch, beta1 = 1, cfg["ot_epoch"], 0.5
    b0 = dd["dataloaders"][tp][0]
    assert b0.graph.n > 9000 and b0.y is not None and b0.y.shape[1] == 9344 and b0.y.dtype == torch.bfloat16
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    staged = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    # (9 281 is not a multiple of 4: the flat buffers hold alignment slots between parameters that no gradient is ever
    # written to -- they are zero in a run and would keep the poison value here -- so the comparison goes parameter by parameter)
    pgrad = lambda: torch.cat([p.grad.reshape(-1) for p in opt.params])
    for rep in range(3):
        for bi in (0, 19):                                        # first and last (partial) batch
            staged.beta1_t[1].fill_(-beta1)
            la = tu.forward_backward(model, cfg, dd, 1, tp, bi, epoch, staged.beta1_t, optimizer=opt)
            ga = pgrad().clone()
            for p in opt.params:
                p.grad.fill_(7.0)
            lb = staged.fb(1, tp, bi, epoch, beta1)
            gb = pgrad()
            assert torch.isfinite(lb).all() and torch.isfinite(gb).all()
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-4, atol=1e-5)
            scale = float(ga.abs().max())
            np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=2e-3, atol=2e-4 * scale)
    assert (la.cpu().numpy()[[1, 3, 4, 5, 6]] > 0).all()
    for k in range(3):
        out = staged.step(1, tp, 0, epoch, beta1)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(opt.flat_param).all()
    del tu, dd, model, opt, staged
    torch.cuda.empty_cache()

    tu, cfg, dd, model, opt = _setup(2, 2000, G, 480, torch.float32)
    b0 = dd["dataloaders"][tp][0]
    noise = sp.make_noise(b0.batch_size, seed=5)
    inp = sp.oracle_inputs(model, dd, cfg, tp, 0, tp - 1)
    w = (cfg["lambda1"], beta1, cfg["beta2"], cfg["omiga1"], cfg["omiga2"], cfg["omiga3"])
    with torch.no_grad():
        _, terms, z_ref = mo.step_loss(inp["P"], inp["svgp"], inp["x"], inp["y"], inp["ei"], inp["b"],
                                       cfg["gat_attention_heads"], noise[0], noise[1], w, km=inp["km"], ot=inp["ot"])
    want = np.array([float(terms[n]) for n in sp.LOSS_NAMES])
    dl, dz, dg = sp.device_step(model, opt, cfg, dd, tu, 1, tp, 0, epoch, beta1, noise)
    np.testing.assert_allclose(dl, want, rtol=1e-4, err_msg=str(sp.LOSS_NAMES))
    np.testing.assert_allclose(dz, z_ref.numpy(), rtol=1e-4, atol=1e-4)
    assert all(np.isfinite(v).all() for v in dg.values())


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_full_size_inference_matches_the_oracle(dtype):
    """SpaDOT.py:96-123 at BASELINE.json's size: all_latent_samples of a whole time point (N_t = 10 000 spots x 3 000 genes) --
    eval-mode BatchNorm (running statistics, moved by a few real training steps first), the SVGP posterior mean over all N_t
    rows, the three GAT layers on the FULL graph -- against oracle/model_oracle.all_latent_samples on the same weights.  This
    runs every epoch for every time point and feeds K-means (_train_utils.py:255-269).  Tolerances: latent relative L2
    <= 1e-5 (fp32) / 3e-3 (bf16); K-means labels of the device latent from given centres = argmin in numpy on the same latent
    (bit-exact), and against the ORACLE latent's labels at most 1 in 10^4 (fp32) / 1 % (bf16) of the spots differ (spots on a
    cluster boundary)."""
    from oracle import model_oracle as mo
    from spadot_amd import ops
    cdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    tu, cfg, dd, model, opt = _setup(2, 10000, 3000, 480, cdt, bf16_exact_inputs=True)
    tp, epoch = 1, cfg["ot_epoch"]
    for k in range(3):                                            # running statistics and weights leave their initial values
        tu.training_step(model, opt, cfg, dd, 1, tp, k, epoch, 0.5)
    torch.cuda.synchronize()
    model.eval()
    loc, Y, _ = dd["datasets"][tp]
    g = dd["graphs"][tp]
    with torch.no_grad():
        lat = model.all_latent_samples(loc, Y, g, tp, as_numpy=False)
    assert lat.shape == (10000, cfg["z_dim"]) and bool(torch.isfinite(lat).all())
    P = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
    assert float(P["SVGPEncoder.SVGP_encoder_net.1.running_mean"].abs().max()) > 0.0
    sv = mo.SVGPOracle(dd["inducing_points"][tp], dd["N_train"][tp], kernel_type=cfg.get("kernel_type", "Gaussian"),
                       scale=cfg.get("kernel_scale", 0.1))
    tgt = torch.repeat_interleave(torch.arange(g.n), (g.rowptr[1:] - g.rowptr[:-1]).cpu().long())
    ei = torch.stack([g.col.cpu().long(), tgt])
    with torch.no_grad():
        ref = mo.all_latent_samples(P, sv, torch.as_tensor(loc).cpu().double(), torch.as_tensor(Y).float().cpu().double(), ei,
                                    cfg["gat_attention_heads"], cfg["z_dim"] // 2, mean_only=True).numpy()
    got = lat.double().cpu().numpy()
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    rel_s = float(np.linalg.norm(got[:, :10] - ref[:, :10]) / np.linalg.norm(ref[:, :10]))
    rel_g = float(np.linalg.norm(got[:, 10:] - ref[:, 10:]) / np.linalg.norm(ref[:, 10:]))
    print(json.dumps({"dtype": dtype, "latent_rel_l2": rel, "svgp_half": rel_s, "gat_half": rel_g}))
    _report(f"inference_full_{dtype}", {"dtype": dtype, "latent_rel_l2": rel, "svgp_half": rel_s, "gat_half": rel_g})
    assert rel <= (3e-3 if dtype == "bf16" else 1e-5), (rel, rel_s, rel_g)
    # K-means assignment from given centres (the ones the setup's refit left behind)
    cen = np.asarray(model.kmeans_center_dict[tp], dtype=np.float64)
    lab_dev = ops.kmeans_assign(lat.double(), torch.as_tensor(cen, device=DEV)).cpu().numpy()
    d_got = np.zeros((got.shape[0], cen.shape[0]))
    for k in range(got.shape[1]):                                   # the kernel's order: dimensions ascending, fp64
        d_got += (got[:, None, k] - cen[None, :, k]) ** 2
    assert np.array_equal(lab_dev, d_got.argmin(1).astype(lab_dev.dtype))
    lab_ref = ((ref[:, None, :] - cen[None]) ** 2).sum(-1).argmin(1)
    frac = float((lab_dev != lab_ref).mean())
    print(json.dumps({"label_mismatch_fraction_vs_oracle_latent": frac}))
    assert frac <= (1e-2 if dtype == "bf16" else 1e-4), frac
