"""GPU: the training driver end to end on a small synthetic data set (plumbing of the whole stage:
prepare_dataloader -> train_SpaDOT -> K-means / OT updates -> outputs), and the per-step glue
(k-means / OT losses) against the reference-generated vectors."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _small_config():
    import yaml
    from spadot_amd.utils import _utils
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(maxiter=3, ot_epoch=1, batch_size=256, inducing_point_nums=60, svgp_encoder_layers=[32, 16],
               gat_encoder_hidden=16, decoder_layers=[16, 32], n_clusters=4, kmeans_backend="sklearn")
    cfg["ot_config"] = dict(cfg["ot_config"], ot_epochs=1)
    return cfg


def test_glue_losses_match_reference_vectors():
    from spadot_amd.utils import _train_utils as tu
    g = load_golden("model_glue.npz")
    model = types.SimpleNamespace(kmeans_center_dict={}, kmeans_cluster_dict={}, kmeans_index_dict={}, gammas={})
    # time point 1 holds 300 spots whose labels are all_labels; the batch is `batch_pos` of them
    all_labels = g["all_labels"]
    tu._set_kmeans_state(model, 1, g["centers1"], all_labels, np.arange(all_labels.size), DEV)
    tu._set_kmeans_state(model, 0, g["centers0"], np.arange(10), np.arange(10), DEV)
    tu._set_gamma(model, "0_1", g["gamma"], DEV)
    # recover the batch positions from the labels stored in the fixture (any positions with these labels do)
    pos = []
    used = set()
    for lab in g["batch_labels"]:
        cand = [i for i in np.nonzero(all_labels == lab)[0] if i not in used][0]
        used.add(cand); pos.append(cand)
    seeds = torch.as_tensor(pos, dtype=torch.int64, device=DEV)
    lat = torch.as_tensor(g["latent"], dtype=torch.float32, device=DEV)
    km = tu._compute_kmeans_loss(model, {}, 1, seeds, lat)
    ot = tu._compute_OT_loss(model, {}, 1, seeds, lat, 0)
    assert float(km) == pytest.approx(float(g["kmeans_loss"]), rel=1e-5)     # fp32 device vs fp64 reference
    assert float(ot) == pytest.approx(float(g["ot_loss"]), rel=1e-5)
    np.testing.assert_array_equal(tu._beta_cycle_linear(100, stop=1.0), g["beta_100_stop1"])


def test_prepare_dataloader_matches_reference_sampling_rules():
    import random
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 1200, 40, seed=5)
    cfg = _small_config()
    cfg.update(input_dim=40, timepoints=[0, 1], device=torch.device(DEV))
    _utils.set_seed(cfg["seed"])
    expected_idx = random.sample(range(2400), cfg["inducing_point_nums"])     # same stream, same call
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    loc = tu._obtain_tp_loc_info(data)
    got = np.concatenate([dd["inducing_points"][0], dd["inducing_points"][1]])
    want = loc[sorted(expected_idx, key=lambda i: (i >= 1200, expected_idx.index(i))), :2]
    np.testing.assert_allclose(got, want)
    assert dd["N_train"] == {0: 1200, 1: 1200}
    # k = min(30, 6*round(1.2)) = 6 neighbours + self loop; batches of 256 seeds, last one partial
    assert dd["graphs"][0].E == 1200 * 7
    assert [b.batch_size for b in dd["dataloaders"][0]] == [256, 256, 256, 256, 176]
    # standardised coordinates per time point
    for tp in (0, 1):
        l = dd["datasets"][tp][0].cpu().numpy()
        np.testing.assert_allclose(l.mean(0), 0, atol=1e-12); np.testing.assert_allclose(l.std(0), 1, rtol=1e-12)


def test_train_end_to_end_writes_reference_outputs(tmp_path):
    import spadot_amd
    from spadot_amd.synthetic import make_dataset
    import yaml
    data = make_dataset(2, 1200, 40, seed=11)
    cfg_path = tmp_path / "cfg.yaml"
    yaml.safe_dump(_small_config(), open(cfg_path, "w"))
    args = types.SimpleNamespace(data=data, output_dir=str(tmp_path / "out"), prefix="t_", config=str(cfg_path),
                                 save_model=True, device=DEV)
    model, loss_df = spadot_amd.train(args)
    out = tmp_path / "out"
    for f in ("t_inducing_points.csv", "loss.csv", "SpaDOT_model.pth", "t_latent.npz"):
        assert (out / f).exists(), f
    import pandas as pd
    loss = pd.read_csv(out / "loss.csv", index_col=0)
    assert list(loss.columns) == ["elbo", "Recon", "SVGP_KL", "GAT_KL", "alignment", "KMeans", "OT"]
    assert loss.shape[0] == 3 and np.isfinite(loss.values).all()
    assert loss["KMeans"].iloc[0] == 0 and loss["KMeans"].iloc[1] > 0     # k-means loss from epoch 1 on
    assert loss["OT"].iloc[0] == 0 and loss["OT"].iloc[1] > 0             # ot_epoch = 1
    assert loss["Recon"].iloc[-1] < loss["Recon"].iloc[0]
    ind = pd.read_csv(out / "t_inducing_points.csv")
    assert list(ind.columns) == ["norm-pixel_x", "norm-pixel_y", "timepoint"] and len(ind) == 60
    z = np.load(out / "t_latent.npz")
    assert z["X"].shape == (2400, 20) and np.isfinite(z["X"]).all()
    sd = torch.load(out / "SpaDOT_model.pth")
    assert "GATEncoder.gat1.lin.weight" in sd and "SVGPEncoder.SVGP_encoder_net.1.running_mean" in sd
    assert set(model.gammas) == {"0_1"} and model.gammas["0_1"].shape == (4, 4)
    assert set(model.kmeans_center_dict) == {0, 1}
    # integer cluster assignments: the device assignment kernel reproduces the labels sklearn's fit left for the
    # same latents (no training step ran between the last _update_Kmeans and get_latent) bit for bit
    from spadot_amd import ops
    for tp in (0, 1):
        lat = torch.as_tensor(z["X"][1200 * tp:1200 * (tp + 1)], device=DEV)        # float32, as the host fit saw it
        cen = torch.as_tensor(np.asarray(model.kmeans_center_dict[tp]), device=DEV)
        got = ops.kmeans_assign(lat, cen).cpu().numpy()
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, np.asarray(model.kmeans_cluster_dict[tp], dtype=np.int32))


def test_graphed_steps_match_eager_steps(monkeypatch):
    """hipGraph replay of the training step == the eager step (reparameterisation noise silenced so that the
    two runs see the same numbers; what differs is only how the launches are issued)."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    monkeypatch.setattr(torch, "randn_like", lambda x, **k: torch.zeros_like(x))
    data = make_dataset(2, 1200, 40, seed=3)
    cfg = _small_config()
    cfg.update(input_dim=40, timepoints=[0, 1], device=torch.device(DEV))
    results = []
    sync = lambda flat: flat.mul_(1.0)                 # stands in for the replicas' all-reduce (identity)
    for graphed in (False, True, "split", "staged"):   # "split": forward+backward graph | gradient exchange | optimizer graph;
        _utils.set_seed(7)                             # "staged": six graphs per step, the two branches on two streams
        cfg["staged_graphs"] = graphed == "staged"
        dd = tu.prepare_dataloader(data, cfg)
        model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
        opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
        tu._update_Kmeans(model, cfg, dd)
        tu._update_OT_matrix(model, cfg)
        model.train()
        stepper = tu.GraphedStepper(model, opt, cfg, dd, grad_sync=sync if graphed == "split" else None) if graphed else None
        losses = []
        for rep in range(4):                       # visit 1 eager, visit 2 capture + replay, visits 3-4 replay
            for bi in range(2):
                if graphed:
                    losses.append(stepper.step(1, 1, bi, 5, 0.3))
                else:
                    losses.append(tu.training_step(model, opt, cfg, dd, 1, 1, bi, 5, 0.3))
        torch.cuda.synchronize()
        results.append((torch.stack(losses).cpu().numpy(), opt.flat_param.detach().cpu().numpy().copy(), int(opt.step_dev.item())))
        if graphed:
            assert len(stepper.graphs) == 2
            assert (stepper.opt_graph is not None and stepper.opt_graph is not False) == (graphed in ("split", "staged"))
    (l0, p0, s0) = results[0]
    for l1, p1, s1 in results[1:]:
        assert s0 == s1 == 8
        np.testing.assert_allclose(l1, l0, rtol=2e-3, atol=1e-4)   # GEMM split-K order / RNG stream differences only
        np.testing.assert_allclose(p1, p0, rtol=0, atol=5e-4)      # 8 AdamW steps of lr 3e-4 each


def test_chained_steps_leave_the_bits_unchained_steps_leave(monkeypatch):
    """GraphedStepper.chained(): the SVGP branch of step k + 1 waits for the FIRST graph of step k's update (gradient norm + the
    SVGP encoder's parameters: FlatAdamW(first=...)) instead of the whole update.  Same launches on the same two streams,
    only one wait is earlier: parameters, BatchNorm statistics and losses must agree with the unchained run and with the
    one-graph update (an optimizer without a `first` group) to the run-to-run noise of the step itself (a few ulp: 1e-7 relative, measured with
    tools/chain_check.py -- the library's small GEMMs are not bit-repeatable); a branch that read the encoder's parameters
    one update late would be off by ~1e-4."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    monkeypatch.setattr(torch, "randn_like", lambda x, **k: torch.zeros_like(x))
    data = make_dataset(2, 1200, 40, seed=3)
    cfg = _small_config()
    cfg.update(input_dim=40, timepoints=[0, 1], device=torch.device(DEV), staged_graphs=True)
    results = []
    for mode in ("one_graph", "unchained", "chained"):
        _utils.set_seed(7)
        dd = tu.prepare_dataloader(data, cfg)
        model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
        if mode == "one_graph":                    # no `first` group: the update is one graph
            opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
            assert opt.head_count == 0
        else:
            opt = FlatAdamW(model.parameters(), lr=cfg["lr"], first=model.SVGPEncoder.parameters())
            assert opt.head_count == sum((p.numel() + 3) // 4 * 4 for p in model.SVGPEncoder.parameters())
        tu._update_Kmeans(model, cfg, dd)
        tu._update_OT_matrix(model, cfg)
        model.train()
        stepper = tu.GraphedStepper(model, opt, cfg, dd)
        losses = []
        import contextlib
        for rep in range(5):                       # visit 1 eager, visit 2 capture + replay, then replays
            with (stepper.chained() if mode == "chained" else contextlib.nullcontext()):
                for bi in range(2):
                    losses.append(stepper.step(1, 1, bi, 5, 0.3).clone())
        torch.cuda.synchronize()
        assert isinstance(stepper.opt_graph, tuple) == (mode != "one_graph")
        bn = torch.cat([b.detach().double().reshape(-1) for b in model.SVGPEncoder.buffers()])
        named = {n: p.detach().clone() for n, p in model.named_parameters()}
        results.append((torch.stack(losses), named, bn, int(opt.step_dev.item())))
    l0, p0, b0, s0 = results[0]
    for l1, p1, b1, s1 in results[1:]:
        assert s0 == s1 == 10
        assert float(((l0 - l1).abs() / (l0.abs() + 1e-6)).max()) < 5e-6
        assert float((b0 - b1).abs().max()) < 2e-6
        for n in p0:
            assert float((p0[n] - p1[n]).abs().max()) < 2e-6, n


def test_flat_backward_with_in_place_weight_gradients_matches_autograd():
    """bf16 compute: the GAT dense maps write their fp32 weight gradient straight into the flat buffer inside
    FlatAdamW.backward; the buffer must equal what plain autograd returns for every parameter, and a plain
    .backward() must not be affected by the in-place path."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 1200, 40, seed=4)
    cfg = _small_config()
    cfg.update(input_dim=40, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16)
    _utils.set_seed(5)
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
    model.train()
    batch = dd["dataloaders"][1][0]
    noise = (torch.zeros((batch.batch_size, 10), device=DEV), torch.zeros((batch.batch_size, 10), device=DEV))

    def loss_fn():
        for m in model.modules():                      # same BatchNorm state for every evaluation
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        rec, skl, gkl, ali, _ = model.forward(batch.x, batch.y, batch.graph, 1, batch.batch_size, noise=noise)
        return 0.1 * rec - 0.5 * skl + 1e-4 * gkl + 0.1 * ali

    ref = torch.autograd.grad(loss_fn(), opt.params, allow_unused=True)
    opt.flat_grad.fill_(123.0)                          # stale content must not survive
    opt.backward(loss_fn())
    for p, g in zip(opt.params, ref):
        want = torch.zeros_like(p) if g is None else g
        np.testing.assert_allclose(p.grad.cpu().numpy(), want.float().cpu().numpy(), rtol=1e-5, atol=1e-7)
    opt.zero_grad()
    loss_fn().backward()                                # accumulate path: once, not twice
    for p, g in zip(opt.params, ref):
        want = torch.zeros_like(p) if g is None else g
        np.testing.assert_allclose(p.grad.cpu().numpy(), want.float().cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_staged_replay_gives_the_single_graph_gradients_bf16(monkeypatch):
    """bf16 compute at a moderately large shape (4000 spots, 1200 genes: multi-workgroup reductions everywhere): the
    flat gradient left by the staged replay (six graphs, two streams) equals the single-graph replay's, parameter by
    parameter, over repeated replays of several batches -- the regression that bit here was a library reduction whose
    memset node misbehaved inside the small graphs from the second replay on."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 4000, 1200, seed=9)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=1200, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16,
               inducing_point_nums=300, n_clusters=6)
    _utils.set_seed(3)
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))   # same (no) noise in both modes
    single = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=False), dd)
    staged = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    assert staged.staged and not single.staged
    for rep in range(4):                                   # eager, capture + replay, replay, replay
        for bi in range(3):
            la = single.fb(1, 1, bi, cfg["ot_epoch"], 0.5)
            ga = opt.flat_grad.clone()
            opt.flat_grad.fill_(7.0)
            lb = staged.fb(1, 1, bi, cfg["ot_epoch"], 0.5)
            gb = opt.flat_grad
            assert torch.isfinite(gb).all()
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-5, atol=1e-6)
            scale = float(ga.abs().max())
            np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=2e-3, atol=2e-4 * scale)


def test_full_size_step_staged_replay_matches_the_eager_step_bf16():
    """BASELINE.json's bench shape for one time-point pair (10 000 spots x 3 000 genes, batches of 512 seeds whose
    two-hop closure is the whole time point, ~240 inducing points, bf16 compute): the flat gradient and the loss
    terms of the staged graph replay equal the eager step's on the same batches, replay after replay."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 10000, 3000, seed=1993)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=3000, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16,
               inducing_point_nums=480)
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    del data
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    b0 = dd["dataloaders"][1][0]
    assert b0.graph.n > 9000 and b0.graph.E > 250000            # the closure really is (almost) the whole time point
    staged = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    ep = cfg["ot_epoch"]
    for rep in range(3):                                         # eager, capture + replay, replay
        for bi in (0, 7):
            staged.beta1_t[1].fill_(-0.5)                        # the weight the stepper sets for beta1 = 0.5
            la = tu.forward_backward(model, cfg, dd, 1, 1, bi, ep, staged.beta1_t, optimizer=opt)
            ga = opt.flat_grad.clone()
            opt.flat_grad.fill_(7.0)
            lb = staged.fb(1, 1, bi, ep, 0.5)
            gb = opt.flat_grad
            assert torch.isfinite(gb).all()
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-4, atol=1e-5)
            scale = float(ga.abs().max())
            np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=2e-3, atol=2e-4 * scale)


def test_full_size_chained_replay_matches_unchained_replay_bf16():
    """The bench shape again (10 000 spots x 3 000 genes, bf16, replayed staged graphs), where the two streams really
    overlap: twelve steps inside GraphedStepper.chained() -- the SVGP branch of a step starts beside the previous step's
    update of everything but its own parameters -- against the same twelve steps unchained, from the same parameters,
    moments, BatchNorm statistics and step count.  Agreement to a few times the step's own run-to-run noise (the same
    sequence run twice: the library's small GEMMs are not bit-repeatable); a branch that read the SVGP encoder one update
    late would differ in the fourth digit."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 10000, 3000, seed=1993)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=3000, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16,
               inducing_point_nums=480, staged_graphs=True)
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    del data
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"], first=model.SVGPEncoder.parameters())
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    st = tu.GraphedStepper(model, opt, cfg, dd)
    ep = cfg["ot_epoch"]
    bufs = [b for b in model.buffers()]
    state0 = (opt.flat_param.clone(), [b.clone() for b in bufs])

    def restore():
        torch.cuda.synchronize()
        opt.flat_param.copy_(state0[0])
        opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_dev.zero_()
        for b, v in zip(bufs, state0[1]):
            b.copy_(v)
        opt.refresh_images()
        torch.cuda.synchronize()

    def run(chained):
        import contextlib
        restore()
        losses = []
        with (st.chained() if chained else contextlib.nullcontext()):
            for k in range(12):
                losses.append(st.step(1, 1, k % 4, ep, 0.5).clone())
        torch.cuda.synchronize()
        return torch.stack(losses).double(), opt.flat_param.detach().double().clone(), torch.cat([b.detach().double().reshape(-1) for b in bufs])

    for _ in range(2):                                          # eager visit, then capture: every graph of the four batches exists
        for bi in range(4):
            st.step(1, 1, bi, ep, 0.5)
    assert isinstance(st.opt_graph, tuple)
    a, a2, b = run(False), run(False), run(True)

    def dist(x, y):
        return (float(((x[0] - y[0]).abs() / (x[0].abs() + 1e-6)).max()), float((x[1] - y[1]).abs().max()), float((x[2] - y[2]).abs().max()))

    noise, diff = dist(a, a2), dist(a, b)
    print("run-to-run", noise, "chained vs unchained", diff)
    assert int(opt.step_dev.item()) == 12
    assert diff[0] <= max(5 * noise[0], 2e-5) and diff[1] <= max(5 * noise[1], 2e-6) and diff[2] <= max(5 * noise[2], 2e-5)


def test_bucketed_exchange_hands_over_final_buckets_without_changing_the_gradient():
    """Optimizer built with the first GAT layer's parameters last + an async exchange hook (what data-parallel ranks on their
    own devices use, round 5): the SAME staged graphs as a single GPU replays -- the queue of deferred gradient work included --
    with `flat_grad[:tail_offset]` handed to the exchange behind the queue (side stream) and the first layer's gradients behind
    the main stream's last backward graph.  The flat gradient must equal the staged replay's without an exchange, bit for bit,
    both buckets must be FINAL when they are handed over, and their order is fixed."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 2000, 300, seed=9)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=300, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16,
               inducing_point_nums=200, n_clusters=6)
    _utils.set_seed(3)
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    first = model.GATEncoder.first_layer_parameters()
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"], last=first, first=model.SVGPEncoder.parameters())
    cut = opt.tail_offset
    assert cut is not None and opt.count - cut >= sum(p.numel() for p in first)
    assert all(p.grad.data_ptr() >= opt.flat_grad.data_ptr() + 4 * cut for p in first)
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    seen = []

    class Handle:
        def wait(self):
            pass

    def fake_async(view):
        # (the clone is enqueued on the stream the exchange is issued on -- where a process group's stream would start waiting)
        seen.append((view.data_ptr() - opt.flat_grad.data_ptr(), view.numel(), view.clone()))
        return Handle()

    plain = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    split = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd, grad_sync=lambda f: f, grad_sync_async=fake_async)
    assert split.overlap and not plain.overlap and split.staged and plain.staged
    for rep in range(4):                                   # eager, capture + replay, replay, replay
        for bi in range(2):
            la = plain.fb(1, 1, bi, cfg["ot_epoch"], 0.5)
            ga = opt.flat_grad.clone()
            opt.flat_grad.fill_(7.0)
            del seen[:]
            lb = split.fb(1, 1, bi, cfg["ot_epoch"], 0.5)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(lb.cpu().numpy(), la.cpu().numpy())
            np.testing.assert_array_equal(opt.flat_grad.cpu().numpy(), ga.cpu().numpy())
            assert [(o_, n_) for o_, n_, _ in seen] == [(0, cut), (4 * cut, opt.count - cut)]
            np.testing.assert_array_equal(seen[0][2].cpu().numpy(), ga[:cut].cpu().numpy())    # final when handed over
            np.testing.assert_array_equal(seen[1][2].cpu().numpy(), ga[cut:].cpu().numpy())
    # a replica without a batch in a step joins the same two collectives in the same order
    del seen[:]
    split.exchange_idle()
    assert [(o_, n_) for o_, n_, _ in seen] == [(0, cut), (4 * cut, opt.count - cut)]


def _dp_worker(rank, world, port, q, staged=None, granularity="batch", sharded=False):
    """One data-parallel rank on cuda:0 (both ranks share the one GPU of the test box; gloo carries the
    collectives, on a real node the backend is nccl = RCCL)."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spadot_amd import parallel as par
        from spadot_amd.synthetic import make_dataset
        from spadot_amd.utils import _train_utils as tu, _utils
        cfg = _small_config()
        cfg.update(maxiter=2, input_dim=40, timepoints=[0, 1, 2], device=torch.device(DEV), shard_granularity=granularity)
        if staged is not None:
            cfg["staged_graphs"] = staged
        if sharded:
            cfg["sharded_update"] = True
        data = make_dataset(3, 1200, 40, seed=11)
        plan = par.configure_shard(data, cfg, world, rank)
        if granularity == "batch":      # 5 batches per time point, 15 units: rank 0 owns the even canonical indices
            assert cfg["owned_timepoints"] == [0, 1, 2]
            assert cfg["owned_batches"][1] == ([1, 3] if rank == 0 else [0, 2, 4])
        else:
            assert cfg["owned_timepoints"] == ([0, 2] if rank == 0 else [1]) and "owned_batches" not in cfg
        _utils.set_seed(cfg["seed"])
        dd = tu.prepare_dataloader(data, cfg)
        if granularity == "batch":
            assert [b is not None for b in dd["dataloaders"][1]] == [bi in cfg["owned_batches"][1] for bi in range(5)]
        model, losses = par.train_SpaDOT_parallel(dd, cfg)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
        q.put((rank, flat, sorted(model.gammas), sorted(model.kmeans_center_dict), np.asarray(losses[1], dtype=np.float64)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("staged,granularity", [(None, "batch"), (True, "batch"), (None, "timepoint")])
def test_data_parallel_training_two_ranks_one_gpu(staged, granularity):
    """staged=None: what two ranks sharing a device get by default (two graphs per step, one exchange in two
    buckets); staged=True: what ranks on their own devices get -- staged graphs with the bucketed exchange issued
    beside the first GAT layer's backward (here forced onto the shared GPU, gloo carrying the async collectives)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, staged, granularity)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, f0, g0, c0, l0), (_, f1, g1, c1, l1) = res
    np.testing.assert_array_equal(f0, f1)              # replicas identical after 2 epochs
    assert g0 == g1 == ["0_1", "1_2"]                  # every rank holds every (sharded) pair plan
    assert c0 == c1 == [0, 1, 2]                       # centres of all time points everywhere
    assert np.isfinite(l0).all() and l0.shape == (7,)
    np.testing.assert_array_equal(l0, l1)              # the epoch's loss record is the reduced one: the same on every rank
    assert l0[6] > 0                                   # OT term live (ot_epoch = 1)


def test_sharded_update_two_ranks_one_gpu_matches_the_all_reduced_update():
    """model_config['sharded_update'] (FlatAdamW.step_sharded / parallel.sharded_update): two ranks, each
    updating half of the flat buffers between a reduce-scatter of the gradient and an all-gather of the parameters (gloo here:
    all-reduce + list all-gather carry the same values), the clip norm from one scalar all-reduce, the bf16 weight images
    refreshed from the gathered parameters.  Replicas identical; and the run ends where the all-reduce + full-update run of
    the same two ranks ends, up to the rounding of the squared norm (summed per slice, then over the ranks)."""
    import socket
    import torch.multiprocessing as mp
    out = {}
    for sharded in (False, True):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, None, "batch", sharded)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        np.testing.assert_array_equal(res[0][1], res[1][1])            # replicas identical
        np.testing.assert_array_equal(res[0][4], res[1][4])
        out[sharded] = res[0]
    scale = np.abs(out[False][1]).max()
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(out[True][4], out[False][4], rtol=2e-3)


def test_device_kmeans_vs_sklearn():
    """Device K-means: sklearn's fit is not pinned bit for bit, so the check is on what both
    must agree on: labels are the nearest centres (exact), the inertia matches sklearn's optimum, the
    partition is the same on well-separated data, and the fit is deterministic."""
    from sklearn.cluster import KMeans
    from sklearn.metrics import adjusted_rand_score
    from spadot_amd.kmeans import KMeansDevice
    from spadot_amd import ops
    rng = np.random.default_rng(0)
    cen = 3.0 * rng.normal(size=(10, 20))
    X = cen[rng.integers(0, 10, 4000)] + 0.5 * rng.normal(size=(4000, 20))
    sk = KMeans(n_clusters=10, random_state=1993, n_init=10).fit(X)
    Xd = torch.as_tensor(X, device=DEV)
    km = KMeansDevice(10, random_state=1993, n_init=10).fit(Xd)
    assert km.labels_.dtype == np.int32 and km.cluster_centers_.shape == (10, 20)
    assert km.inertia_ == pytest.approx(sk.inertia_, rel=1e-6)
    assert adjusted_rand_score(sk.labels_, km.labels_) > 0.999
    d = ((X[:, None, :] - km.cluster_centers_[None]) ** 2).sum(-1)
    np.testing.assert_array_equal(km.labels_, d.argmin(1).astype(np.int32))       # integer labels: exact rule
    km2 = KMeansDevice(10, random_state=1993, n_init=10).fit(Xd)
    np.testing.assert_array_equal(km.labels_, km2.labels_)
    np.testing.assert_array_equal(km.cluster_centers_, km2.cluster_centers_)
    # harder, overlapping data: inertia within 1 % of sklearn's best of 10
    Xh = rng.normal(size=(3000, 20)) + 0.8 * cen[rng.integers(0, 10, 3000)] / 3.0
    skh = KMeans(n_clusters=10, random_state=1993, n_init=10).fit(Xh)
    kmh = KMeansDevice(10, random_state=1993, n_init=10).fit(torch.as_tensor(Xh, device=DEV))
    assert kmh.inertia_ <= 1.01 * skh.inertia_


def test_device_kmeans_fit_many_matches_fit_per_data_set():
    """spadot_amd.kmeans.fit_many (the per-epoch K-means of ALL time points as one batched fit: _update_Kmeans) against
    KMeansDevice.fit on each data set alone, ragged sizes: the same partition and inertia (centres to rounding: batched
    products may round differently), labels = the exact nearest-centre rule, deterministic."""
    from sklearn.metrics import adjusted_rand_score
    from spadot_amd.kmeans import KMeansDevice, fit_many
    rng = np.random.default_rng(4)
    Xs = []
    for n in (1500, 700, 2300, 257):
        cen = 3.0 * rng.normal(size=(10, 20))
        Xs.append(torch.as_tensor(cen[rng.integers(0, 10, n)] + 0.5 * rng.normal(size=(n, 20)), device=DEV))
    many = fit_many(Xs, 10, random_state=1993, n_init=10)
    again = fit_many(Xs, 10, random_state=1993, n_init=10)
    assert len(many) == 4
    for X, km, km2 in zip(Xs, many, again):
        one = KMeansDevice(10, random_state=1993, n_init=10).fit(X)
        assert km.labels_.dtype == np.int32 and km.labels_.shape == (X.shape[0],) and km.cluster_centers_.shape == (10, 20)
        assert km.inertia_ == pytest.approx(one.inertia_, rel=1e-9)
        assert adjusted_rand_score(one.labels_, km.labels_) > 0.9999
        order = np.argsort(km.cluster_centers_[:, 0]); order1 = np.argsort(one.cluster_centers_[:, 0])
        np.testing.assert_allclose(km.cluster_centers_[order], one.cluster_centers_[order1], rtol=1e-9, atol=1e-9)
        Xh = X.cpu().numpy()
        d = ((Xh[:, None, :] - km.cluster_centers_[None]) ** 2).sum(-1)
        np.testing.assert_array_equal(km.labels_, d.argmin(1).astype(np.int32))
        np.testing.assert_array_equal(km.labels_, km2.labels_)
        np.testing.assert_array_equal(km.cluster_centers_, km2.cluster_centers_)


@pytest.mark.parametrize("n_ind", [1100, 1500])
def test_training_with_more_inducing_points_than_one_sweep_takes(tmp_path, capsys, n_ind):
    """The default inducing-point load (1200 over two time points -> m ~ 600 each; here ~550 and ~750) is beyond one
    sweep launch: the blocked elimination keeps the step capturable, graphs are replayed, losses stay finite."""
    import spadot_amd, yaml
    from spadot_amd.synthetic import make_dataset
    cfg = _small_config(); cfg["inducing_point_nums"] = n_ind; cfg["maxiter"] = 3
    p = tmp_path / "cfg.yaml"; yaml.safe_dump(cfg, open(p, "w"))
    args = types.SimpleNamespace(data=make_dataset(2, 1200, 40, seed=11), output_dir=str(tmp_path / "o"), prefix="",
                                 config=str(p), save_model=False, device=DEV)
    model, loss = spadot_amd.train(args)
    assert np.isfinite(loss.values).all()
    m = max(int(s.inducing_index_points.shape[0]) for s in model.svgp_dict.values())
    assert m > 310
    assert "run eagerly" not in capsys.readouterr().out


def test_training_with_device_kmeans_and_knn_backends(tmp_path):
    """Everything after the h5ad read on the device: spatial kNN graph, K-means fits, training (bf16 compute)."""
    import spadot_amd, yaml
    from spadot_amd.synthetic import make_dataset
    cfg = _small_config(); cfg["kmeans_backend"] = "device"; cfg["knn_backend"] = "device"; cfg["maxiter"] = 2
    p = tmp_path / "cfg.yaml"; yaml.safe_dump(cfg, open(p, "w"))
    args = types.SimpleNamespace(data=make_dataset(2, 1200, 40, seed=11), output_dir=str(tmp_path / "o"), prefix="",
                                 config=str(p), save_model=False, device=DEV)
    model, loss = spadot_amd.train(args)
    assert np.isfinite(loss.values).all() and set(model.kmeans_center_dict) == {0, 1}
    assert len(model.kmeans_cluster_dict[0]) == 1200


def test_cfg4_shape_eight_ragged_timepoints_train_end_to_end(tmp_path, capsys):
    """BASELINE.json configs[3] shape on one GPU (MouseOrganogenesis-like: 8 time points, here with ragged small spot
    counts): whole epochs through spadot_amd.train -- every (time point, batch) step as replayed graphs, all 8 K-means
    refits and all 7 consecutive-pair OT solves (one launch of the small solver) per epoch
    (_train_utils.py:174-231, 309-321)."""
    import spadot_amd
    import yaml
    from spadot_amd.synthetic import make_dataset
    counts = [500, 700, 650, 900, 800, 550, 600, 750]
    data = make_dataset(8, counts, 40, seed=3)
    cfg = _small_config()
    cfg.update(maxiter=4, inducing_point_nums=160, kmeans_backend="device")
    cfg_path = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    args = types.SimpleNamespace(data=data, output_dir=str(tmp_path / "out"), prefix="m_", config=str(cfg_path),
                                 save_model=False, device=DEV)
    model, loss_df = spadot_amd.train(args)
    out = capsys.readouterr().out
    assert out.count("OT iter 0") == 7 * 4                       # 7 pair solves after each of the 4 epochs (ot_epochs = 1)
    loss = loss_df.T
    assert loss.shape == (4, 7) and np.isfinite(loss.values).all()
    assert (loss["KMeans"].iloc[1:] > 0).all() and (loss["OT"].iloc[1:] > 0).all()
    assert set(model.kmeans_center_dict) == set(range(8))
    assert sorted(model.gammas) == sorted(f"{t}_{t + 1}" for t in range(7))
    for t in range(8):
        assert np.asarray(model.kmeans_center_dict[t]).shape == (4, 20)
        assert len(model.kmeans_cluster_dict[t]) == counts[t]
    for key, g in model.gammas.items():
        assert g.shape == (4, 4) and np.isfinite(g).all() and g.sum() > 0
        dev = model._gamma_dev[key].cpu().numpy()               # what the OT loss kernel reads: rows normalised
        np.testing.assert_allclose(dev, g / g.sum(axis=1, keepdims=True), rtol=2e-6)
    z = np.load(tmp_path / "out" / "m_latent.npz")
    assert z["X"].shape == (sum(counts), 20) and np.isfinite(z["X"]).all()
    steps = sum(-(-c // 256) for c in counts)
    assert steps == 25


def test_weight_images_follow_every_writer_of_the_weights():
    """the GAT encoder skips its per-step bf16 weight cast while an optimizer keeps the images current.
    The fast path must validate itself: (1) a SECOND FlatAdamW over the same model (the eager path after a stepper) re-points
    the parameters -- the first optimizer no longer owns them, the pin is dropped and the forward casts again; (2) a write
    the optimizer did not make (load_state_dict, p.copy_, flat_param.copy_) is caught in front of the next replayed step.
    Checked on the images themselves: after each event + one forward/step the images equal bf16(weights)."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 1500, 256, seed=11)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=256, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16,
               inducing_point_nums=120, n_clusters=4, kmeans_backend="sklearn")
    _utils.set_seed(5)
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"], first=model.SVGPEncoder.parameters())
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    enc = model.GATEncoder
    st = tu.GraphedStepper(model, opt, cfg, dd)
    ep = cfg["ot_epoch"]

    def images_current(after_update=True):
        """max |image - bf16(weight)| over the three layers (the images the NEXT forward would read)."""
        from spadot_amd.ops import weight_image
        torch.cuda.synchronize()
        worst = 0.0
        for layer, K in ((enc.gat1, 256), (enc.gat2, enc.gat2.in_channels), (enc.gat3, enc.gat2.in_channels)):
            W = layer.lin.weight.detach()
            im = weight_image(layer.lin.weight, K, torch.bfloat16, layer)
            assert bool(torch.isfinite(W).all()), "weights went non-finite: the test's perturbations are too large"
            worst = max(worst, float((im[:, :W.shape[1]].float() - W.to(torch.bfloat16).float()).abs().max()))
        return worst

    for _ in range(3):                                    # eager, capture, replay: the optimizer now maintains the images
        for bi in range(2):
            st.step(1, 1, bi, ep, 0.5)
    # (the three GAT layers' weights and the decoder's output map: decoder.py _output_image)
    assert enc._image_optimizer is opt and model.decoder._image_optimizer is opt and len(opt._images) == 4
    assert images_current() == 0.0

    def output_image_error():
        from spadot_amd.ops import weight_image
        torch.cuda.synchronize()
        W = list(model.decoder.decoder_net)[-1].weight
        im = weight_image(W, W.shape[1], torch.bfloat16, model.decoder, tag="_wout")
        return float((im.float() - W.detach().to(torch.bfloat16).float()).abs().max())

    assert output_image_error() == 0.0
    with torch.no_grad():
        list(model.decoder.decoder_net)[-1].weight.mul_(1.04)      # a write the optimizer did not make: caught by step()'s sync
    assert output_image_error() > 0.0
    st.step(1, 1, 0, ep, 0.5)
    assert output_image_error() == 0.0

    # (2a) load_state_dict between two replayed steps
    sd = {k: (v.clone() * 1.03 if k.endswith("lin.weight") else v.clone()) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    assert images_current() == 0.0                         # the post hook of whichever optimizer is pinned
    # (2b) a write through the parameter and one through the flat buffer: caught by step()'s sync
    with torch.no_grad():
        enc.gat2.lin.weight.mul_(1.05)
    assert images_current() > 0.0
    st.step(1, 1, 0, ep, 0.5)
    assert images_current() == 0.0
    opt.flat_param.mul_(0.97)
    assert images_current() > 0.0
    st.step(1, 1, 1, ep, 0.5)
    assert images_current() == 0.0

    # (1) a second optimizer + the eager step on the same model
    opt2 = FlatAdamW(model.parameters(), lr=cfg["lr"])
    assert not opt.owns(enc.gat1.lin.weight) and opt2.owns(enc.gat1.lin.weight)
    before = enc.gat1.lin.weight.detach().clone()
    for bi in range(2):
        tu.training_step(model, opt2, cfg, dd, 1, 1, bi, ep, 0.5)
    torch.cuda.synchronize()
    assert enc._image_optimizer is None and model.decoder._image_optimizer is None      # the stale pins are gone
    assert float((enc.gat1.lin.weight.detach() - before).abs().max()) > 0.0
    # the eager path casts the images at the head of every forward (nobody maintains them now): behind the last update
    # they lag, the next forward brings them up to date
    batch = dd["dataloaders"][1][0]
    assert batch.y is not None and images_current() > 0.0
    with torch.no_grad():
        enc.pre_head(batch.y, batch.graph, rows=batch.batch_size)
    assert images_current() == 0.0
    # a maintain_image() on the optimizer that lost the parameters is refused
    from spadot_amd.ops import weight_image
    assert not opt.maintain_image(enc.gat1.lin.weight, weight_image(enc.gat1.lin.weight, 256, torch.bfloat16, enc.gat1))


def test_small_timepoint_bf16_staged_replay_matches_eager():
    """A time point with fewer than 1 024 spots in the bf16 compute dtype (ChickenHeart's first: 747): the weight-gradient and
    dense-map kernels refuse such row counts and the LIBRARY products run instead -- the deferred ones inside the `late`
    queue, i.e. outside any autograd backward.  Round 5's ChickenHeart-shaped run found that such a product written with out=
    into a flat-gradient view from a saved activation that requires grad turned the whole flat buffer into a non-leaf (the
    queue now runs without a tape).  Staged replay == eager step, flat buffer still a plain leaf, replay after replay; G is
    not a multiple of 4 either (2 954 genes there)."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, [747, 1966], 602, seed=4)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=602, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16, inducing_point_nums=300)
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"], first=model.SVGPEncoder.parameters())
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    staged = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    ep = cfg["ot_epoch"]
    pgrad = lambda: torch.cat([p.grad.reshape(-1) for p in opt.params])
    for rep in range(3):
        for tp_i, tp, bi in ((0, 0, 1), (1, 1, 3), (0, 0, 0)):          # partial last batches (235, 430 seeds) and a full one
            staged.beta1_t[1].fill_(-0.5)
            la = tu.forward_backward(model, cfg, dd, tp_i, tp, bi, ep, staged.beta1_t, optimizer=opt)
            ga = pgrad().clone()
            for p in opt.params:
                p.grad.fill_(7.0)
            lb = staged.fb(tp_i, tp, bi, ep, 0.5)
            torch.cuda.synchronize()
            assert not opt.flat_grad.requires_grad and all(not p.grad.requires_grad for p in opt.params)
            gb = pgrad()
            assert torch.isfinite(lb).all() and torch.isfinite(gb).all()
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=2e-3, atol=2e-4 * float(ga.abs().max()))
    for k in range(4):
        out = staged.step(k % 2, k % 2, k % 2, ep, 0.5)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(opt.flat_param).all()


def test_deferred_weight_gradients_leave_the_same_gradient():
    """The staged replay (the default arrangement): the second GAT layer's weight gradient and attention / bias sums, the last
    layer's weight / attention-vector chain and the decoder output map's weight gradient are queued by the backward functions
    and run later on the side stream; the SVGP backward's gradient-independent half is formed beside the tail with q1 through
    T = X2 S K_mn, the posterior is handed over before the ELBO scalars, the cluster terms and their gradient are one launch.
    Against the SINGLE-GRAPH replay of the same batches (`staged_graphs: false`: the plain step body, nothing queued, nothing
    precomputed, the reconstruction stage as GEMM + kernel + GEMM): other routes through the same algebra, so the comparison is
    to rounding (1e-4 of the largest gradient entry beyond the single-graph replay's own run-to-run noise), not bit for bit -- bf16, 4000 spots x 1200 genes: the matrix-core
    paths and the aggregate-first last layer."""
    from spadot_amd.model import SpaDOT
    from spadot_amd.ops import FlatAdamW
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    data = make_dataset(2, 4000, 1200, seed=9)
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(input_dim=1200, timepoints=[0, 1], device=torch.device(DEV), compute_dtype=torch.bfloat16,
               inducing_point_nums=300, n_clusters=6)
    _utils.set_seed(3)
    dd = tu.prepare_dataloader(data, cfg)
    model = SpaDOT.SpaDOT(cfg, dd).to(DEV)
    opt = FlatAdamW(model.parameters(), lr=cfg["lr"], first=model.SVGPEncoder.parameters())
    tu._update_Kmeans(model, cfg, dd)
    tu._update_OT_matrix(model, cfg)
    model.train()
    model.fixed_noise = (torch.zeros((512, 10), device=DEV), torch.zeros((512, 10), device=DEV))
    on = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=True), dd)
    off = tu.GraphedStepper(model, opt, dict(cfg, staged_graphs=False), dd)
    assert on.staged and not off.staged
    assert model.GATEncoder.gat2.defer_wgrad            # (the flag on the layer only permits queueing: nothing queues in `off`)
    ep = cfg["ot_epoch"]
    for rep in range(3):                                   # eager, capture + replay, replay
        for bi in (0, 2):
            la = off.fb(1, 1, bi, ep, 0.5).clone()
            ga = opt.flat_grad.clone()
            off.fb(1, 1, bi, ep, 0.5)
            ga2 = opt.flat_grad.clone()                    # the same replay again: the library's run-to-run noise, if any
            opt.flat_grad.fill_(7.0)                       # poison: a queued write that never lands stays visible
            lb = on.fb(1, 1, bi, ep, 0.5)
            torch.cuda.synchronize()
            scale = float(ga.abs().max())
            noise = float((ga2 - ga).abs().max())
            # the tolerance below is built from the replay's own run-to-run difference: bound THAT first, so that a racy replay
            # cannot widen its own tolerance (the library's split-K GEMMs are the only non-bit-repeatable launches of the step)
            assert noise <= 1e-6 * scale, (noise, scale)
            diff = float((opt.flat_grad - ga).abs().max())
            # (1e-7 until the reconstruction stage became one launch: its output map accumulates in another order than the
            # library's, which flips the bf16 rounding of a few entries of d recon / d o -- 2^-9 of an entry each -- and moves
            # the gradients that are sums over them by ~1e-5 of their scale; measured 0.8e-5 and 2.6e-5 on the two batches.  A queued write
            # that is lost or lands on the wrong operand is off by the scale itself: the poison above makes that visible)
            assert diff <= 4.0 * noise + 1e-4 * scale, (diff, noise)
            np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # the gradients that travel through the queue are really there: the last layer's, the second layer's weight gradient, and
    # (round 5) the attention-vector / bias sums of the two matrix-core layers -- the second layer's through the queue, the first
    # layer's at the end of its backward stage, both written straight into the flat buffer
    g1, g2, g3 = model.GATEncoder.gat1, model.GATEncoder.gat2, model.GATEncoder.gat3
    assert float(g3.att_src.grad.abs().max()) > 0 and float(g3.lin.weight.grad.abs().max()) > 0
    assert float(g2.lin.weight.grad.abs().max()) > 0
    for layer in (g1, g2):
        for p_ in (layer.att_src, layer.att_dst, layer.bias):
            assert 0 < float(p_.grad.abs().max()) < 7.0
            assert p_.grad.data_ptr() >= opt.flat_grad.data_ptr()
