"""GPU: the matrix-core GAT edge kernels (csrc/gat_mfma.hip, bf16 rows, block plans) against
  (a) the per-edge kernels of model_kernels.hip on the same inputs (what they replace: same arithmetic, other order), and
  (b) an independent fp64 scatter formulation of GATConv's message passing.
Tolerances: outputs are bf16 (8 significant bits, rtol 2^-7 on a value, the same rounding in both HIP paths); the weights
keep 16 bits (bf16 hi + lo), accumulation is fp32 -- so the two HIP paths agree to bf16 output rounding, and gradients of
parameters (sums over all nodes) to ~1e-2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from spadot_amd import ops
    return ops


def _problem(n, k, hub, seed, n_tgt=None, spatial=True):
    from spadot_amd.graph import attach_plans, build_batch_graph, knn_graph, morton_key
    rng = np.random.default_rng(seed)
    coords = rng.uniform(size=(n, 2)) * np.sqrt(n)
    ei = knn_graph(coords, k)
    if hub:        # node 5 hears from everybody: in-degree n (a column list far longer than one chunk)
        extra = np.stack([np.arange(n), np.full(n, 5)])
        ei = np.unique(np.concatenate([ei, extra], axis=1), axis=1)
    if n_tgt is not None:                    # only the first n_tgt nodes are targets (layer-2 shape)
        ei = ei[:, ei[1] < n_tgt]
        from spadot_amd.graph import _csr_both
        from spadot_amd.ops import BatchGraph
        src, dst = ei[0], ei[1]
        keep = src != dst
        src = np.concatenate([src[keep], np.arange(n_tgt)]); dst = np.concatenate([dst[keep], np.arange(n_tgt)])
        g = BatchGraph(n, *(torch.from_numpy(np.ascontiguousarray(p)).to(DEV) for p in _csr_both(src, dst, n, n_tgt)), n_tgt=n_tgt)
    else:
        g = build_batch_graph(ei, n, DEV)
    attach_plans(g, morton_key(coords) if spatial else None)
    assert g.plan_t is not None and g.plan_s is not None
    return g


def _inputs(g, H, C, seed):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    h = (torch.randn((g.n, H * C), device=DEV, generator=gen) * 0.5).bfloat16()
    a_s = torch.randn((1, H, C), device=DEV, generator=gen) * 0.1
    a_d = torch.randn((1, H, C), device=DEV, generator=gen) * 0.1
    bias = 0.1 * torch.randn(H * C, device=DEV, generator=gen)
    w = torch.randn((g.n_tgt, H * C), device=DEV, generator=gen).bfloat16()
    return h, a_s, a_d, bias, w


def _run(ops, g, h, a_s, a_d, bias, w, H, C, act, mfma):
    ops.GAT_MFMA[0] = mfma
    try:
        t = [x.clone().requires_grad_(True) for x in (h, a_s, a_d, bias)]
        out = ops.gat_edge(t[0], t[1], t[2], t[3], g, H, C, True, act)
        (out.float() * w.float()).sum().backward()
        return out.detach(), [x.grad.detach().float() for x in t]
    finally:
        ops.GAT_MFMA[0] = True


@pytest.mark.parametrize("n,k,H,C,act,hub,n_tgt,spatial", [
    (97, 7, 4, 512, True, False, None, True),          # fewer rows than one block of 32 x 4
    (300, 9, 2, 512, False, True, None, True),         # a hub: 300 columns in one block (in-degree 300: 5 softmax passes)
    (1000, 30, 4, 512, True, False, None, True),       # the layer shape
    (1000, 30, 4, 512, True, False, 400, True),        # layer-2 shape: targets are a prefix of the nodes
    (500, 12, 8, 512, False, False, None, False),      # no spatial order: long column lists, many chunks; 8 heads
    (33, 4, 1, 512, True, False, None, True),          # one full block + one row; a single head
])
def test_mfma_path_matches_per_edge_kernels(ops, n, k, H, C, act, hub, n_tgt, spatial):
    g = _problem(n, k, hub, seed=n + C, n_tgt=n_tgt, spatial=spatial)
    h, a_s, a_d, bias, w = _inputs(g, H, C, seed=3)
    assert ops._mfma_plans(h, g, H, C, True) is not None
    out_m, gr_m = _run(ops, g, h, a_s, a_d, bias, w, H, C, act, True)
    out_e, gr_e = _run(ops, g, h, a_s, a_d, bias, w, H, C, act, False)
    assert torch.isfinite(out_m.float()).all()
    # forward: same bf16 rounding of the same fp32 sums (other summation order): at most one bf16 ulp apart
    np.testing.assert_allclose(out_m.float().cpu().numpy(), out_e.float().cpu().numpy(), rtol=2 ** -7, atol=1e-3)
    # dh is bf16 too; att / bias gradients are fp32 sums over all nodes of bf16-rounded terms
    for name, a, b in zip(("h", "att_src", "att_dst", "bias"), gr_m, gr_e):
        a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
        scale = np.abs(b).max() + 1e-30
        if name == "h":
            # element-wise, except where an output within rounding of 0 took the other LeakyReLU slope in one of the two
            # summation orders (act = True: a handful of elements out of millions, each worth one term of one row)
            bad = np.abs(a - b) > 2 ** -6 * np.abs(b) + 4e-3 * scale
            assert bad.sum() <= max(2, 1e-5 * bad.size) if act else not bad.any(), (name, int(bad.sum()), np.abs(a - b).max())
            assert np.linalg.norm(a - b) <= 5e-3 * np.linalg.norm(b), (name, np.linalg.norm(a - b) / np.linalg.norm(b))
        else:
            assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b) + 1e-6, (name, np.linalg.norm(a - b), np.linalg.norm(b))
    # deterministic: no atomics, fixed order
    out_m2, gr_m2 = _run(ops, g, h, a_s, a_d, bias, w, H, C, act, True)
    assert torch.equal(out_m, out_m2) and all(torch.equal(a, b) for a, b in zip(gr_m, gr_m2))


def test_mfma_path_full_size_vs_fp64_scatter(ops):
    """cfg3 layer shape (10k nodes, k = 30 + self loops, H = 4, C = 512, bf16 rows) against the fp64 scatter formulation
    evaluated on the SAME bf16-rounded inputs: what is checked is the kernels' own error (bf16 output rounding), not
    the input quantisation."""
    from spadot_amd.graph import attach_plans, build_batch_graph, knn_graph, morton_key
    rng = np.random.default_rng(0)
    n, H, C, k = 10000, 4, 512, 30
    side = int(np.sqrt(n))
    coords = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2) + rng.uniform(-0.3, 0.3, (n, 2))
    coords = coords[rng.permutation(n)]
    g = attach_plans(build_batch_graph(knn_graph(coords, k), n, DEV), morton_key(coords))
    assert g.E == n * (k + 1) and g.plan_t.avg_cols < 200
    h, a_s, a_d, bias, w = _inputs(g, H, C, seed=1)
    out, grads = _run(ops, g, h, a_s, a_d, bias, w, H, C, True, True)
    f64 = torch.float64
    hd, s1d, s2d, bd = (t.detach().to(f64).requires_grad_(True) for t in (h, a_s, a_d, bias))
    tgt = torch.repeat_interleave(torch.arange(n, device=DEV), (g.rowptr[1:] - g.rowptr[:-1]).long())
    src = g.col.long()
    hv = hd.view(n, H, C)
    pre = (hv * s1d).sum(-1)[src] + (hv * s2d).sum(-1)[tgt]
    e = torch.nn.functional.leaky_relu(pre, 0.2)
    emax = torch.full((n, H), -float("inf"), device=DEV, dtype=f64).scatter_reduce(0, tgt[:, None].expand(-1, H), e, "amax")
    ex = torch.exp(e - emax[tgt])
    den = torch.zeros((n, H), device=DEV, dtype=f64).index_add_(0, tgt, ex) + 1e-16
    alpha = ex / den[tgt]
    opre = torch.zeros((n, H, C), device=DEV, dtype=f64).index_add_(0, tgt, alpha[:, :, None] * hv[src]).reshape(n, H * C) + bd
    ref = torch.nn.functional.leaky_relu(opre, 0.01)
    (ref * w.to(f64)).sum().backward()
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2 ** -8, atol=1e-4)
    for name, a, t in zip(("h", "att_src", "att_dst", "bias"), grads, (hd, s1d, s2d, bd)):
        r, a = t.grad.cpu().numpy(), a.cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(a - r) / np.linalg.norm(r)
        assert rel <= (6e-3 if name == "h" else 2e-2), (name, rel)
