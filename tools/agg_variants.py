"""k_gat_agg<1> (source-side product of the GAT backward) with and without the folded extras, cfg3 batch shape:
lean (ds_src given) | + ds_src summed by the workgroups from dz | + attention-vector partials | both."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd import ops, _lib
from spadot_amd.graph import knn_graph, precompute_batches
from spadot_amd.ops import _p, _stream, DT_BF16
dev = "cuda"
rng = np.random.default_rng(0)
n, k, H, C = 10000, 30, 4, 512
side = int(np.sqrt(n))
coords = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2) + rng.uniform(-0.3, 0.3, (n, 2))
coords = coords[rng.permutation(n)]
ei = knn_graph(coords, k)
g = precompute_batches(ei, n, 512, dev, coords=coords, plans=True)[3].graph
ps = g.plan_s
lib = _lib.model_lib()
nn = g.n
gp = (torch.randn((nn, H * C), device=dev) * 0.5).bfloat16()
h = (torch.randn((nn, H * C), device=dev) * 0.5).bfloat16()
img = ps.weight_image(H)
img.copy_((torch.rand_like(img.float()) * 0.03).to(img.dtype))
a_s = torch.randn(H * C, device=dev); a_d = torch.randn(H * C, device=dev)
ds_src = torch.randn((nn, H), device=dev); ds_dst = torch.randn((nn, H), device=dev)
dz = torch.randn((g.E, H), device=dev)
dh = torch.empty_like(h)
W3 = 3 * H * C
part = torch.zeros((ps.nb, W3), device=dev)
def run(with_dz, with_att):
    return lib.spadot_gat_aggregate(_p(gp), DT_BF16, _p(img), _p(ps.rows), _p(ps.sptr), _p(ps.cols), ps.nb, ps.max_cols, H, C, 1,
                                    _p(a_s), _p(a_d), 0, None if with_dz else _p(ds_src), _p(ds_dst), _p(dh),
                                    _p(h) if with_att else None, _p(part) if with_att else None, W3, nn, nn,
                                    _p(dz) if with_dz else None, _p(g.rowptr_t) if with_dz else None,
                                    _p(g.eid_t) if with_dz else None, None, _stream())
for rnd in range(2):
    for name, a, b in (("lean", False, False), ("+ds from dz", True, False), ("+att partials", False, True), ("both", True, True)):
        for _ in range(10): assert run(a, b) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run(a, b)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:16s} {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us", flush=True)
