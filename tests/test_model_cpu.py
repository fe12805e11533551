"""CPU: host-side model logic that needs no kernel launch -- state_dict layout, graph/batch construction."""
import numpy as np
import torch

from conftest import load_golden
from oracle import model_oracle as mo


def _cfg(g, device="cpu"):
    return dict(input_dim=g["Y"].shape[1], z_dim=20, device=device, svgp_encoder_layers=[24, 12],
                gat_encoder_hidden=8, gat_attention_heads=int(g["heads"]), decoder_layers=[12, 24],
                kernel_type="Gaussian", kernel_scale=0.1, timepoints=[0, 1])


def test_state_dict_keys_and_shapes_match_reference():
    from spadot_amd.model import SpaDOT
    g = load_golden("model_composite.npz")
    m = SpaDOT.SpaDOT(_cfg(g), {"inducing_points": {0: g["ind0"], 1: g["ind1"]},
                                "N_train": {0: float(g["N_train0"]), 1: float(g["N_train1"])}})
    ref = {k[3:]: g[k].shape for k in g.files if k.startswith("sd/")}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(mine) == set(ref)
    for k in ref:
        assert mine[k] == tuple(ref[k]), k
    # the reference's fp64 checkpoint loads (values are cast to this path's fp32 parameters)
    m.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd/")})


def test_knn_graph_matches_oracle_and_reference_edge_counts():
    from spadot_amd.graph import knn_graph
    rng = np.random.default_rng(0)
    for n, k in [(150, 6), (747, 6), (500, 12)]:
        c = rng.uniform(0, 30, size=(n, 2))
        ei = knn_graph(c, k)
        np.testing.assert_array_equal(ei, mo.knn_graph(c, k).numpy())
        assert ei.shape[1] == n * (k + 1)          # n*k directed edges + n self loops
    # ChickenHeart notebook: 747 spots, k=6 -> 4482 kNN edges (examples/ChickenHeart.ipynb:221-230)
    assert knn_graph(rng.uniform(size=(747, 2)), 6).shape[1] - 747 == 4482


def test_induced_batches_match_oracle_and_fixture():
    from spadot_amd.graph import induced_batch
    g = load_golden("model_composite.npz")
    n = g["X"].shape[0]
    n_id, sub = induced_batch(g["edge_index"], n, np.arange(int(g["batch_size"])))
    np.testing.assert_array_equal(n_id, g["n_id"])
    np.testing.assert_array_equal(sub, g["sub_edge_index"])
    # a local (grid-ordered) graph gives a batch much smaller than the time point
    side = 40
    xy = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2).astype(float)
    from spadot_amd.graph import knn_graph
    ei = knn_graph(xy + 0.01 * np.random.default_rng(1).normal(size=xy.shape), 6)
    n_id2, sub2 = induced_batch(ei, side * side, np.arange(64))
    o_id, o_sub = mo.induced_batch(torch.as_tensor(ei), side * side, np.arange(64))
    np.testing.assert_array_equal(n_id2, o_id.numpy())
    np.testing.assert_array_equal(sub2, o_sub.numpy())
    assert 64 < n_id2.size < side * side // 2


def test_csr_round_trip():
    from spadot_amd.graph import _csr_both
    rng = np.random.default_rng(2)
    n = 50
    src = rng.integers(0, n, 400); dst = rng.integers(0, n, 400)
    keep = src != dst
    src = np.concatenate([src[keep], np.arange(n)]); dst = np.concatenate([dst[keep], np.arange(n)])
    rowptr, col, rowptr_t, col_t, eid_t = _csr_both(src, dst, n)
    # target-ordered edges
    tgt = np.repeat(np.arange(n), np.diff(rowptr))
    assert sorted(zip(col.tolist(), tgt.tolist())) == sorted(zip(src.tolist(), dst.tolist()))
    # transposed view enumerates the same edges, eid_t points back into the target order
    s_t = np.repeat(np.arange(n), np.diff(rowptr_t))
    np.testing.assert_array_equal(col[eid_t], s_t)
    np.testing.assert_array_equal(tgt[eid_t], col_t)


def test_gatconv_initialisation_bounds_follow_pyg_glorot():
    """torch_geometric's GATConv.reset_parameters: glorot(att_src / att_dst) draws U(-s, s) with
    s = sqrt(6 / (size(-2) + size(-1))) = sqrt(6 / (H + C)) on the [1, H, C] tensors, and the
    reference re-draws lin.weight Xavier-uniform (encoder.py:42-46): bound sqrt(6 / (fan_in + fan_out))."""
    from spadot_amd.model.encoder import GATConv
    torch.manual_seed(0)
    H, C, fin = 4, 512, 300
    layer = GATConv(fin, C, heads=H, concat=True)
    s_att = (6.0 / (H + C)) ** 0.5
    for att in (layer.att_src, layer.att_dst):
        a = att.detach()
        assert a.shape == (1, H, C)
        assert float(a.abs().max()) <= s_att
        assert float(a.abs().max()) > 0.95 * s_att            # 2048 draws: the maximum sits next to the bound
        assert abs(float(a.std()) - s_att / 3 ** 0.5) < 0.05 * s_att
    s_lin = (6.0 / (fin + H * C)) ** 0.5
    w = layer.lin.weight.detach()
    assert float(w.abs().max()) <= s_lin and float(w.abs().max()) > 0.99 * s_lin
    assert float(layer.bias.abs().max()) == 0.0


def test_config_yaml_is_the_reference_schema_plus_three_documented_keys():
    """The packaged config.yaml carries every key of the reference's SpaDOT/config.yaml with the same default
    (values pinned here from /root/reference/SpaDOT/config.yaml:1-57, which does not travel) and exactly three keys of
    its own; `compute_dtype` strings resolve to torch dtypes."""
    import types
    from spadot_amd.utils import _utils
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    ref = dict(maxiter=100, ot_epoch=50, batch_size=512, z_dim=20, n_clusters=10, seed=1993, lr=3e-4,
               svgp_encoder_layers=[256, 64], gat_encoder_hidden=512, gat_attention_heads=4, decoder_layers=[64, 256],
               kernel_type="Gaussian", kernel_scale=0.1, inducing_point_nums=1200, lambda1=0.1, beta1=1.0, beta2=1e-4,
               knn_cutoff=6, max_neighbors=30, omiga1=0.1, omiga2=0.1, omiga3=1.0)
    for k, v in ref.items():
        assert cfg[k] == v, k
    ot = dict(growth_iters=3, ot_epochs=10, epsilon=0.05, epsilon0=1, lambda1=0.1, lambda2=5.0, tau=1000, scaling_iter=3000,
              inner_iter_max=50, tolerance=1e-8, max_iter=10000000, batch_size=5, extra_iter=1000, numItermax=1000000,
              use_Py=False, use_C=True, profiling=False, method="waddington")
    assert cfg["ot_config"] == ot
    assert set(cfg) - set(ref) - {"ot_config"} == {"compute_dtype", "kmeans_backend", "knn_backend"}
    assert (cfg["compute_dtype"], cfg["kmeans_backend"], cfg["knn_backend"]) == ("float32", "device", "sklearn")
    assert _utils.resolve_compute_dtype(cfg["compute_dtype"]) is torch.float32
    assert _utils.resolve_compute_dtype("bfloat16") is torch.bfloat16 and _utils.resolve_compute_dtype("bf16") is torch.bfloat16
    assert _utils.resolve_compute_dtype(None) is torch.float32 and _utils.resolve_compute_dtype(torch.bfloat16) is torch.bfloat16
    import pytest
    with pytest.raises(ValueError):
        _utils.resolve_compute_dtype("float16")


def test_decoder_hidden_gradients_form_one_block_of_a_flat_buffer():
    """ops._contiguous_grad_block: when the hidden stages' (W, bias, gamma, beta) gradient views sit back to back in one flat
    buffer in module order (what FlatAdamW lays out on the device), they are returned as ONE tensor over that storage -- what
    lets the fused decoder backward (csrc/mlp_chain.hip) write its result row straight into it.  Host logic only."""
    import torch
    from spadot_amd import ops
    from spadot_amd.model.decoder import Decoder
    dec = Decoder(input_dim=40, z_dim=20, decoder_layers=[64, 256])
    plist = list(dec.parameters())
    flat = torch.zeros(sum(p.numel() for p in plist))
    off = 0
    for p in plist:                                   # FlatAdamW's layout: module order, sizes here are multiples of 4
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    stages = list(dec.decoder_net)
    params = [t for i in range(0, len(stages) - 1, 3) for t in (stages[i].weight, stages[i].bias, stages[i + 1].weight, stages[i + 1].bias)]
    blk = ops._contiguous_grad_block(params)
    assert blk is not None and blk.numel() == sum(p.numel() for p in params)
    blk.fill_(3.0)
    assert all(float(p.grad.min()) == 3.0 and float(p.grad.max()) == 3.0 for p in params)
    assert float(stages[-1].weight.grad.abs().max()) == 0.0            # the output map's slot is not part of the block
    assert ops._contiguous_grad_block(params[::-1]) is None            # another order is not one block
    stages[0].bias.grad = torch.zeros(64)                              # a gradient outside the buffer breaks the block
    assert ops._contiguous_grad_block(params) is None
