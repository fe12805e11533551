"""Diagnostic: full-size GAT edge kernels vs a torch scatter formulation (fp32 and fp64) for several seeds."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from spadot_amd import ops
from spadot_amd.graph import knn_graph, build_batch_graph
DEV = "cuda"
n, H, C, k = 10000, 4, 512, 30
rng = np.random.default_rng(0)
g = build_batch_graph(knn_graph(rng.uniform(size=(n, 2)), k), n, DEV)
tgt = torch.repeat_interleave(torch.arange(n, device=DEV), (g.rowptr[1:] - g.rowptr[:-1]).long())
src = g.col.long()

def ref(h, s1, s2, bias, w, dt):
    h, s1, s2, bias, w = (t.detach().to(dt).requires_grad_(True) for t in (h, s1, s2, bias, w))
    hv = h.view(n, H, C)
    pre = (hv * s1).sum(-1)[src] + (hv * s2).sum(-1)[tgt]
    e = torch.nn.functional.leaky_relu(pre, 0.2)
    emax = torch.full((n, H), -float("inf"), device=DEV, dtype=dt).scatter_reduce(0, tgt[:, None].expand(-1, H), e, "amax")
    ex = torch.exp(e - emax[tgt])
    den = torch.zeros((n, H), device=DEV, dtype=dt).index_add_(0, tgt, ex) + 1e-16
    alpha = ex / den[tgt]
    o = torch.zeros((n, H, C), device=DEV, dtype=dt).index_add_(0, tgt, alpha[:, :, None] * hv[src])
    opre = o.reshape(n, H * C) + bias
    o = torch.nn.functional.leaky_relu(opre, 0.01)
    (o * w.detach()).sum().backward()
    return o.detach(), h.grad, s1.grad, s2.grad, bias.grad, pre.detach(), opre.detach()

for seed in range(6):
    torch.manual_seed(seed)
    h = (torch.randn((n, H * C), device=DEV) * 0.5).requires_grad_(True)
    s1 = (torch.randn((1, H, C), device=DEV) * 0.1).requires_grad_(True)
    s2 = (torch.randn((1, H, C), device=DEV) * 0.1).requires_grad_(True)
    bias = (0.1 * torch.randn(H * C, device=DEV)).requires_grad_(True)
    out = ops.gat_edge(h, s1, s2, bias, g, H, C, True, True)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    r32 = ref(h, s1, s2, bias, w, torch.float32)
    r64 = ref(h, s1, s2, bias, w, torch.float64)
    gh64 = r64[1]
    mx = gh64.abs().max()
    for nm, a in (("hip", h.grad), ("t32", r32[1])):
        d = (a.double() - gh64).abs()
        bad = d > 2e-3 * gh64.abs() + 2e-4 * mx
        rows = bad.any(1).sum().item()
        print(seed, nm, "bad", int(bad.sum()), "rows", rows, "maxerr/max", float(d.max() / mx))
    pre = r64[5]
    print("   logit |pre|<1e-6:", int((pre.abs() < 1e-6).sum()), "<1e-5:", int((pre.abs() < 1e-5).sum()),
          " out |pre|<1e-7:", int((r64[6].abs() < 1e-7).sum()))
