"""GPU: the fused hidden stages of the decoder (csrc/mlp_chain.hip) against a torch fp64 reference of the same stages
([Linear, LayerNorm, LeakyReLU] x n, /root/reference/SpaDOT/model/decoder.py:3-20) and against the per-stage launches they
replace.  fp32 kernels, fp32 accumulation: values within 2e-5 relative to the fp64 reference, parameter gradients (sums over
the batch) within 1e-4 of their norm."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stages(dims, seed):
    torch.manual_seed(seed)
    out = []
    for din, dout in zip(dims[:-1], dims[1:]):
        lin, ln = nn.Linear(din, dout), nn.LayerNorm(dout)
        with torch.no_grad():
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.uniform_(-0.3, 0.3)
        out.append((lin.to(DEV), ln.to(DEV), 0.01 if len(out) % 2 == 0 else 0.2))
    return out


def _reference(x, stages, w):
    h = x.double()
    ps = []
    for lin, ln, slope in stages:
        W, b, g, be = (t.detach().double().requires_grad_(True) for t in (lin.weight, lin.bias, ln.weight, ln.bias))
        ps += [W, b, g, be]
        h = torch.nn.functional.leaky_relu(torch.nn.functional.layer_norm(h @ W.t() + b, (W.shape[0],), g, be, ln.eps), slope)
    (h * w.double()).sum().backward()
    return h.detach(), [p.grad for p in ps]


def _run(ops, x, stages, w, fused):
    ops.MLP_CHAIN[0] = fused
    try:
        for lin, ln, _ in stages:
            for p in (*lin.parameters(), *ln.parameters()):
                p.grad = None
        xin = x.clone().requires_grad_(True)
        if fused:
            assert ops.mlp_chain_ok(xin, stages)
            h = ops.mlp_chain(xin, stages)
        else:
            h = xin
            for lin, ln, slope in stages:
                h = ops.ln_act(ops.linear_bias(h, lin.weight, lin.bias), ln, slope)
        (h * w).sum().backward()
        grads = [p.grad.detach().clone() for lin, ln, _ in stages for p in (lin.weight, lin.bias, ln.weight, ln.bias)]
        return h.detach(), xin.grad.detach(), grads
    finally:
        ops.MLP_CHAIN[0] = True


@pytest.mark.parametrize("b,dims", [
    (1024, [20, 64, 256]),        # the decoder of config.yaml at cfg3's batch
    (37, [20, 64, 256]),          # fewer rows than one backward block of 32 + a remainder
    (1000, [8, 16]),              # a single stage
    (513, [12, 24, 8, 256]),      # three stages, widths going down and up, one row past a block
    (64, [256, 256]),             # the widest stage the kernel takes
])
def test_chain_matches_fp64_reference_and_unfused_path(b, dims):
    from spadot_amd import ops
    stages = _stages(dims, seed=b)
    g = torch.Generator(device=DEV).manual_seed(b + 1)
    x = torch.randn((b, dims[0]), device=DEV, generator=g)
    w = torch.randn((b, dims[-1]), device=DEV, generator=g)
    xr = x.double().requires_grad_(True)
    ref, ref_grads = _reference(xr, stages, w)
    out, dx, grads = _run(ops, x, stages, w, True)
    out_u, dx_u, grads_u = _run(ops, x, stages, w, False)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out.cpu().numpy(), out_u.cpu().numpy(), rtol=2e-5, atol=2e-5)
    rdx = xr.grad
    assert (dx.double() - rdx).norm() <= 1e-4 * rdx.norm() + 1e-9
    assert (dx - dx_u).norm() <= 1e-4 * dx_u.norm() + 1e-9
    for k, (a, r, u) in enumerate(zip(grads, ref_grads, grads_u)):
        assert a.shape == r.shape
        assert (a.double() - r).norm() <= 1e-4 * r.norm() + 1e-7, (k, float((a.double() - r).norm() / r.norm()))
        assert (a - u).norm() <= 2e-4 * u.norm() + 1e-7, k
    # fixed summation order: the same bits on every launch
    out2, dx2, grads2 = _run(ops, x, stages, w, True)
    assert torch.equal(out, out2) and torch.equal(dx, dx2) and all(torch.equal(p, q) for p, q in zip(grads, grads2))


def test_chain_declines_what_it_does_not_cover():
    from spadot_amd import ops
    x = torch.randn((16, 6), device=DEV)
    assert not ops.mlp_chain_ok(x, _stages([6, 16], 0))            # input width not a multiple of 4
    x = torch.randn((16, 8), device=DEV)
    assert not ops.mlp_chain_ok(x, _stages([8, 12], 0))            # output width not a multiple of 8
    assert not ops.mlp_chain_ok(x, _stages([8, 512], 0))           # wider than the kernel's LDS rows
    assert ops.mlp_chain_ok(x, _stages([8, 16], 0))


def test_decoder_uses_the_chain_and_keeps_its_gradients():
    """Decoder.forward with and without the fused stages: same reconstruction, same parameter gradients."""
    from spadot_amd import ops
    from spadot_amd.model.decoder import Decoder
    torch.manual_seed(3)
    dec = Decoder(input_dim=300, z_dim=20, decoder_layers=[64, 256]).to(DEV)
    z = torch.randn((512, 20), device=DEV)
    res = []
    for fused in (True, False):
        ops.MLP_CHAIN[0] = fused
        try:
            dec.zero_grad(set_to_none=True)
            out = dec(z)
            out.square().mean().backward()
            res.append((out.detach().clone(), [p.grad.detach().clone() for p in dec.parameters()]))
        finally:
            ops.MLP_CHAIN[0] = True
    np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), rtol=1e-4, atol=1e-5)
    for a, u in zip(res[0][1], res[1][1]):
        assert (a - u).norm() <= 2e-4 * u.norm() + 1e-8


@pytest.mark.parametrize("b,G,K", [(1024, 3000, 256), (37, 100, 64), (512, 33, 8)])
def test_fused_reconstruction_term_matches_the_separate_launches(b, G, K):
    """Decoder.recon_loss in the bf16 compute dtype: [cast, GEMM, bias + squared error + sum] against linear_bias + sqerr_sum
    (the same bf16 GEMM, so the values agree to summation order), and against an fp64 evaluation of the same bf16 operands."""
    from spadot_amd import ops
    g = torch.Generator(device=DEV).manual_seed(G)
    h = torch.randn((b, K), device=DEV, generator=g)
    W = (torch.randn((G, K), device=DEV, generator=g) * 0.1)
    bias = torch.randn(G, device=DEV, generator=g) * 0.1
    y = torch.randn((b, G), device=DEV, generator=g)
    inv = 1.0 / G
    res = []
    for fused in (True, False):
        hh, WW, bb = (t.clone().requires_grad_(True) for t in (h, W, bias))
        if fused:
            assert ops.recon_sqerr_ok(hh, WW, bb, y)
            loss = ops.recon_sqerr(hh, WW, bb, y, inv)
        else:
            loss = ops.sqerr_sum(y, ops.linear_bias(hh, WW, bb, torch.bfloat16), inv)
        (loss * 0.37).backward()
        res.append((loss.detach().double().item(), hh.grad, WW.grad, bb.grad))
    ref = inv * ((y.double() - (h.bfloat16().double() @ W.bfloat16().double().t() + bias.double())) ** 2).sum().item()
    assert abs(res[0][0] - ref) <= 2e-6 * abs(ref)
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(ref)
    for a, u in zip(res[0][1:], res[1][1:]):
        assert (a - u).norm() <= 1e-5 * u.norm() + 1e-9
    # same bits on every launch
    hh, WW, bb = (t.clone().requires_grad_(True) for t in (h, W, bias))
    loss = ops.recon_sqerr(hh, WW, bb, y, inv)
    (loss * 0.37).backward()
    assert loss.item() == res[0][0] or abs(loss.double().item() - res[0][0]) == 0.0
    assert torch.equal(bb.grad, res[0][3])


@pytest.mark.parametrize("b,K,N", [(512, 512, 20), (37, 64, 20), (1000, 512, 32), (8, 8, 1)])
def test_head_fc_from_bf16_rows_matches_float_linear(b, K, N):
    """ops.head_fc (GAT_fc on the bf16 rows of the last GAT layer) against linear_bias on h.float(): the same fp32 products of
    the same bf16-rounded inputs, other summation order; dh is rounded to bf16 once."""
    from spadot_amd import ops
    g = torch.Generator(device=DEV).manual_seed(K + N)
    h = torch.randn((b, K), device=DEV, generator=g).bfloat16()
    W = torch.randn((N, K), device=DEV, generator=g) * 0.05
    bias = torch.randn(N, device=DEV, generator=g) * 0.1
    w = torch.randn((b, N), device=DEV, generator=g)
    res = []
    for fused in (True, False):
        hh = h.clone().requires_grad_(True)
        WW, bb = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        if fused:
            assert ops.head_fc_ok(hh, WW, bb)
            out = ops.head_fc(hh, WW, bb)
        else:
            out = ops.linear_bias(hh.float(), WW, bb)
        (out * w).sum().backward()
        res.append((out.detach(), hh.grad, WW.grad, bb.grad))
    ref = h.double() @ W.double().t() + bias.double()
    np.testing.assert_allclose(res[0][0].cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert res[0][1].dtype == torch.bfloat16
    dh_ref = (w.double() @ W.double())
    assert (res[0][1].double() - dh_ref).norm() <= 3e-3 * dh_ref.norm() + 1e-9          # one bf16 rounding per element
    for a, u in zip(res[0][2:], res[1][2:]):
        assert (a - u).norm() <= 1e-5 * u.norm() + 1e-7
    hh = h.clone().requires_grad_(True)
    WW, bb = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    out = ops.head_fc(hh, WW, bb)
    (out * w).sum().backward()
    assert torch.equal(out.detach(), res[0][0]) and torch.equal(WW.grad, res[0][2]) and torch.equal(hh.grad, res[0][1])


@pytest.mark.parametrize("b,dims", [(512, [20, 64, 256]), (37, [12, 24, 8, 256])])
def test_chain_bf16_copy_and_added_input_gradient(b, dims):
    """mlp_chain(bf16_out=True): the same launch leaves the bf16 rounding of its result; mlp_chain(dx_add=e): the backward launch
    returns dx + e -- both without touching any other output (bit-identical values and parameter gradients)."""
    from spadot_amd import ops
    stages = _stages(dims, 11)
    g = torch.Generator(device=DEV).manual_seed(b)
    x = torch.randn((b, dims[0]), device=DEV, generator=g)
    w = torch.randn((b, dims[-1]), device=DEV, generator=g)
    extra = torch.randn((b, dims[0]), device=DEV, generator=g)
    h0, dx0, g0 = _run(ops, x, stages, w, True)
    for lin, ln, _ in stages:
        for p in (*lin.parameters(), *ln.parameters()):
            p.grad = None
    xin = x.clone().requires_grad_(True)
    h, hb = ops.mlp_chain(xin, stages, bf16_out=True, dx_add=extra)
    (h * w).sum().backward()
    assert torch.equal(h.detach(), h0) and hb.dtype == torch.bfloat16 and torch.equal(hb, h0.bfloat16())
    assert torch.equal(xin.grad, dx0 + extra)
    for a, u in zip([p.grad for lin, ln, _ in stages for p in (lin.weight, lin.bias, ln.weight, ln.bias)], g0):
        assert torch.equal(a, u)
    # the fallback for consumers without such a launch: an identity whose backward adds the constant
    xin = x.clone().requires_grad_(True)
    (ops.grad_bias(xin, extra) * 2.0).sum().backward()
    assert torch.equal(xin.grad, torch.full_like(x, 2.0) + extra)


def test_reconstruction_term_from_ready_bf16_operands():
    """recon_sqerr with the bf16 copy of h (mlp_chain's) and a current bf16 image of W handed in: no cast launch, same bits."""
    from spadot_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    b, G, K = 512, 3000, 256
    h = torch.randn((b, K), device=DEV, generator=g)
    W = torch.randn((G, K), device=DEV, generator=g) * 0.1
    bias = torch.randn(G, device=DEV, generator=g) * 0.1
    y = torch.randn((b, G), device=DEV, generator=g)
    res = []
    for ready in (False, True):
        hh, WW, bb = (t.clone().requires_grad_(True) for t in (h, W, bias))
        loss = ops.recon_sqerr(hh, WW, bb, y, 1.0 / G, h.bfloat16() if ready else None, W.bfloat16() if ready else None)
        loss.backward()
        res.append((loss.detach().clone(), hh.grad, WW.grad, bb.grad))
    for a, u in zip(res[0], res[1]):
        assert torch.equal(a, u)
    # operands of the wrong shape or dtype are ignored (cast path), not trusted
    hh, WW, bb = (t.clone().requires_grad_(True) for t in (h, W, bias))
    loss = ops.recon_sqerr(hh, WW, bb, y, 1.0 / G, h[:, :8].bfloat16().contiguous(), W.half())
    assert torch.equal(loss.detach(), res[0][0])


@pytest.mark.parametrize("b,G", [(512, 3000), (430, 2000), (100, 136), (235, 5000)])
def test_recon_forward_and_backward_in_one_launch_match_the_two_gemm_path(b, G):
    """ops.recon_sqerr_fb (csrc/recon_fb.hip, round 5): the decoder's output map, the reconstruction term and their backward for
    a seed known at forward time, one launch on the matrix cores, against ops.recon_sqerr (library GEMM, bias + squared error,
    library GEMM) on the same bf16 operands and against fp64: value, d/dh, dW, dbias; partial row and gene blocks; bit-repeatable;
    a backward seeded with anything but the promised scalar is refused."""
    from spadot_amd import ops
    K = 256
    g = torch.Generator(device=DEV).manual_seed(b + G)
    h = torch.randn((b, K), device=DEV, generator=g)
    hb = h.bfloat16()
    W = torch.randn((G, K), device=DEV, generator=g) * 0.06
    Wb = W.bfloat16()
    bias = torch.randn(G, device=DEV, generator=g) * 0.1
    y = torch.randn((b, G), device=DEV, generator=g)
    inv = 1.0 / G
    weights = torch.tensor([0.1, -0.5, 1e-4, 0.1, 0.1, 1.0], device=DEV)
    gw = weights[0]
    WT = torch.zeros((K, (G + 127) // 128 * 128), device=DEV, dtype=torch.bfloat16)
    WT[:, :G] = Wb.t()
    assert ops.recon_fb_ok(h, W, bias, y, hb, Wb, WT, gw)

    def run_fb():
        hh, WW, bb = h.clone().requires_grad_(True), W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        out = ops.recon_sqerr_fb(hh, WW, bb, y, inv, hb, Wb, WT, gw)
        dh, dW, db = torch.autograd.grad(out, [hh, WW, bb], grad_outputs=weights[0])
        return out.detach().clone(), dh.clone(), dW.clone(), db.clone()

    hh, WW, bb = h.clone().requires_grad_(True), W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = ops.recon_sqerr(hh, WW, bb, y, inv, hb, Wb)
    rdh, rdW, rdb = torch.autograd.grad(ref, [hh, WW, bb], grad_outputs=weights[0])
    out, dh, dW, db = run_fb()
    torch.cuda.synchronize()
    # fp64 from the same bf16-rounded operands
    o64 = hb.double() @ Wb.double().t() + bias.double()
    d64 = y.double() - o64
    assert abs(float(out) - float((d64 ** 2).sum() * inv)) <= 1e-5 * float((d64 ** 2).sum() * inv)
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    g64 = -2.0 * inv * 0.1 * d64
    np.testing.assert_allclose(db.cpu().numpy(), g64.sum(0).cpu().numpy(), rtol=5e-3, atol=5e-3 * float(g64.sum(0).abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), rdb.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(rdb.abs().max()) + 1e-9)
    sc = float(rdh.abs().max())
    np.testing.assert_allclose(dh.cpu().numpy(), rdh.cpu().numpy(), rtol=2e-3, atol=2e-3 * sc)       # (g is rounded to bf16 in both paths)
    np.testing.assert_allclose(dh.cpu().numpy(), (g64 @ Wb.double()).cpu().numpy(), rtol=2e-2, atol=1e-2 * sc)
    np.testing.assert_allclose(dW.cpu().numpy(), rdW.cpu().numpy(), rtol=2e-3, atol=2e-3 * float(rdW.abs().max()))
    out2, dh2, dW2, db2 = run_fb()
    assert torch.equal(out, out2) and torch.equal(dh, dh2) and torch.equal(db, db2)
    # another seed is refused
    hh = h.clone().requires_grad_(True)
    out3 = ops.recon_sqerr_fb(hh, W.clone().requires_grad_(True), bias.clone().requires_grad_(True), y, inv, hb, Wb, WT, gw)
    with pytest.raises(RuntimeError):
        torch.autograd.grad(out3, [hh], grad_outputs=torch.tensor(0.1, device=DEV))
